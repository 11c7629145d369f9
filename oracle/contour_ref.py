"""ORACLE (test infrastructure, never imported by the product path): label image -> per-instance outer contour polygons.

CPU restatement of what the reference does after inference on the OMERO upload route:
  * ``get_indices_pandas`` (/root/reference/src/utils/hull_polygon.py:8-41): label image -> {id: (rows, cols)} pixel lists;
  * ``cv2_countour`` (hull_polygon.py:44-89): the pixel list is painted into its bounding box + 1 px margin and handed to
    ``cv2.findContours(mask, cv2.RETR_TREE, cv2.CHAIN_APPROX_NONE)``; for an instance without holes the single contour,
    and for an instance with holes the contour that ``covers`` the others (its outer border), is returned as a (2, N)
    array [rows; cols] in image coordinates;
  * the points string of an OMERO polygon ROI, ``"x,y x,y ... "`` (src/inference/infer.py:283-287).

PARITY UNPINNED: the algorithm lives in a third-party dependency that is absent here — OpenCV (``opencv=4.5.3``,
requirements.yml) and shapely are installed in neither interpreter of this container, and the reference has no test or
vector for this path — so no golden vector can be generated.  This file restates the published algorithm that
``cv2.findContours`` implements (Suzuki & Abe 1985, "Topological structural analysis of digitized binary images by border
following", outer-border case; OpenCV 4.5 modules/imgproc/src/contours.cpp ``icvFetchContour``):

  direction codes 0..7 = E, NE, N, NW, W, SW, S, SE (image coordinates, y down);
  start = first pixel of the component in raster order (its west neighbour is background);
  1. from direction W turn CLOCKWISE (NW, N, NE, E, SE, S, SW) to the first foreground neighbour i1; none: 1-pixel contour;
  2. at the current pixel i3 (first: the start), with s = direction towards the previous pixel (first: towards i1), turn
     COUNTER-CLOCKWISE from s + 1 to the first foreground neighbour i4; emit i3; stop when i3 == i1 and i4 == start, else
     move to i4.
  Every visited pixel is emitted (CHAIN_APPROX_NONE), pixels of 1-px-wide parts twice; foreground is 8-connected.

tests/test_polygons.py anchors it on known answers (the 3 x 3 square of the OpenCV documentation order: top-left first,
then DOWN the left edge; single pixel; line; diagonal; plus-shape) and on properties (every emitted point is a border
pixel of the instance, consecutive points are 8-neighbours, the polygon's filled interior contains the instance).
"""
import numpy as np

# (dy, dx) per direction code: E, NE, N, NW, W, SW, S, SE
DELTAS = ((0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1))


def get_indices(data, background_id=0):
    """{mask id: (rows, cols)} in ascending id order, pixels of an id in raster order (hull_polygon.py:8-41)."""
    data = np.asarray(data)
    rows, cols = np.nonzero(data != background_id)
    ids = data[rows, cols]
    order = np.argsort(ids, kind="stable")
    rows, cols, ids = rows[order], cols[order], ids[order]
    cuts = np.flatnonzero(np.diff(ids)) + 1
    return {int(i[0]): (r, c) for i, r, c in zip(np.split(ids, cuts), np.split(rows, cuts), np.split(cols, cuts))} \
        if len(ids) else {}


def trace_outer_border(fg, start):
    """fg: 2-D bool array; start: (row, col) of the raster-first pixel of an 8-connected component.
    Returns the list of (row, col) border points in OpenCV's order."""
    H, W = fg.shape

    def on(y, x):
        return 0 <= y < H and 0 <= x < W and bool(fg[y, x])

    y0, x0 = start
    s = 4
    while True:                                   # step 1: clockwise from W
        s = (s - 1) & 7
        if s == 4 or on(y0 + DELTAS[s][0], x0 + DELTAS[s][1]):
            break
    if s == 4:
        return [(y0, x0)]
    i1 = (y0 + DELTAS[s][0], x0 + DELTAS[s][1])
    pts = []
    y, x = y0, x0
    while True:                                   # step 2: counter-clockwise from s + 1
        for k in range(1, 9):
            d = (s + k) & 7
            ny, nx = y + DELTAS[d][0], x + DELTAS[d][1]
            if on(ny, nx):
                break
        pts.append((y, x))
        if (ny, nx) == (y0, x0) and (y, x) == i1:
            return pts
        y, x = ny, nx
        s = (d + 4) & 7


def label_polygons(labels, background_id=0):
    """{id: [(2, N) int array [rows; cols], ...]}: one outer contour per 8-connected component of each id, components in
    raster order of their first pixel (watershed instances have exactly one)."""
    labels = np.asarray(labels)
    out = {}
    for i, (rows, cols) in get_indices(labels, background_id).items():
        fg = labels == i
        seen = np.zeros(labels.shape, bool)
        polys = []
        for r, c in zip(rows, cols):              # raster order
            if seen[r, c]:
                continue
            comp = _component(fg, (r, c))
            seen |= comp
            pts = trace_outer_border(comp, (r, c))
            polys.append(np.array(pts, dtype=np.int64).T.reshape(2, -1))
        out[i] = polys
    return out


def _component(fg, start):
    """8-connected component of ``start`` in ``fg`` (bool image)"""
    comp = np.zeros(fg.shape, bool)
    stack = [start]
    comp[start] = True
    H, W = fg.shape
    while stack:
        y, x = stack.pop()
        for dy, dx in DELTAS:
            ny, nx = y + dy, x + dx
            if 0 <= ny < H and 0 <= nx < W and fg[ny, nx] and not comp[ny, nx]:
                comp[ny, nx] = True
                stack.append((ny, nx))
    return comp


def points_string(polygon):
    """(2, N) [rows; cols] -> the reference's OMERO polygon points string "x,y x,y ... " (infer.py:283-286)."""
    return "".join("{},{} ".format(polygon[1, k], polygon[0, k]) for k in range(polygon.shape[1]))
