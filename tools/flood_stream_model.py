"""The marker phase of skimage's watershed on a CONSTANT image (the boundary method: watershed(image=mask, ...), reference
postprocessing.py:88) without the heap — the derivation that csrc/postproc.hip::pp_flood_const_stream_kernel implements.

`ref_order`: the textbook binary heap of the reference (oracle/postproc_ref.c heap_push / heap_pop, keys (value, age), all
markers age 0).  `model_order`: the same pop and push sequence from three O(1) rules —
  * the heap array starts as the markers in raster order = an implicit complete binary tree; markers surface in its
    PREORDER (a sinking pushed entry shifts the first marker-holding path up by one),
  * array positions turn into pushed entries in its POSTORDER,
  * a marker still sitting in the array's last slot ties with the root's children, stays at the root and pops next ("jumps").
`python tools/flood_stream_model.py [n]` sweeps random masks / markers (densities 0 .. 1, blobs, stripes, single pixels, all
pixels) and compares the two pop for pop; tests/test_flood_model.py runs a part of it on the CPU."""
import numpy as np, sys

def ref_order(mask, markers):
    H, W = mask.shape
    out = np.where(mask, markers, 0).astype(np.int64).ravel()
    m = mask.ravel()
    heap = []   # list of (age, idx) — value constant
    def less(a, b): return a[0] < b[0]
    def push(e):
        heap.append(e); c = len(heap) - 1
        while c > 0:
            p = (c + 1) // 2 - 1
            if less(heap[c], heap[p]): heap[c], heap[p] = heap[p], heap[c]; c = p
            else: break
    def pop():
        top = heap[0]; last = heap.pop()
        n = len(heap)
        if n == 0: return top
        heap[0] = last; i = 0
        while True:
            l, r = 2 * i + 1, 2 * i + 2
            s = i
            if l < n:
                if less(heap[l], heap[i]): s = l
                if r < n and less(heap[r], heap[s]): s = r
            else: break
            if s == i: break
            heap[i], heap[s] = heap[s], heap[i]; i = s
        return top
    for i in np.nonzero(out)[0]: push((0, int(i)))
    age = 0; order = []; pushes = []
    while heap:
        a, idx = pop()
        order.append(idx)
        y, x = divmod(idx, W)
        for ny, nx in ((y - 1, x), (y, x - 1), (y, x + 1), (y + 1, x)):
            if ny < 0 or ny >= H or nx < 0 or nx >= W: continue
            j = ny * W + nx
            if not m[j] or out[j] != 0: continue
            age += 1; out[j] = out[idx]; push((age, j)); pushes.append(j)
    return order, out.reshape(H, W), pushes

def preorder_next(p, n):
    """next position after p in the preorder of the implicit heap tree of n nodes, or -1"""
    l = 2 * p + 1
    if l < n: return l
    while p > 0:
        if p % 2 == 1 and p + 1 < n: return p + 1      # left child with a right sibling
        p = (p - 1) // 2
    return -1

def postorder_first(n):
    p = 0
    while 2 * p + 1 < n: p = 2 * p + 1
    return p

def postorder_next(p, n):
    if p == 0: return -1
    if p % 2 == 1 and p + 1 < n:                          # left child: go to the right sibling's first
        q = p + 1
        while 2 * q + 1 < n: q = 2 * q + 1
        return q
    return (p - 1) // 2

def model_order(mask, markers):
    """phase 1 only (the markers), then returns what the heap-free model predicts: order of marker pops, labels after
    phase 1, pushes in age order"""
    H, W = mask.shape
    out = np.where(mask, markers, 0).astype(np.int64).ravel()
    m = mask.ravel()
    mk = [int(i) for i in np.nonzero(out)[0]]          # raster order = heap array order
    n = len(mk)
    order = []; pushes = []
    if n == 0: return order, out.reshape(H, W), pushes
    count = n; zn = n
    jumped = np.zeros(n, bool); conv = np.zeros(n, bool)
    pre = 0                                            # preorder cursor: position whose marker is at the root now
    post = postorder_first(n)
    cur = 0                                            # position (original) of the marker at the root
    npop = 0
    while True:
        # pop: output the root's marker
        idx = mk[cur]; order.append(idx); npop += 1
        if npop == n: 
            # remaining heap entries are pushed ones; process this marker's neighbours and stop
            pass
        count -= 1
        nxt = None
        if npop < n:
            tailpos = count
            if tailpos < zn and not conv[tailpos] and tailpos != 0 and not jumped[tailpos] and count > 0:
                # the tail is an original marker still in place: it jumps to the root
                zn = tailpos
                jumped[tailpos] = True
                nxt = tailpos
            else:
                if tailpos < zn: zn = tailpos
                # a pushed entry sinks: one position converts (postorder), the preorder-next marker surfaces
                while post != -1 and (post >= zn or conv[post] or jumped[post]): post = postorder_next(post, n)
                if post != -1: conv[post] = True
                pre = preorder_next(pre, n)
                while pre != -1 and jumped[pre]: pre = preorder_next(pre, n)
                nxt = pre
        y, x = divmod(idx, W)
        for ny, nx in ((y - 1, x), (y, x - 1), (y, x + 1), (y + 1, x)):
            if ny < 0 or ny >= H or nx < 0 or nx >= W: continue
            j = ny * W + nx
            if not m[j] or out[j] != 0: continue
            out[j] = out[idx]; pushes.append(j); count += 1
        if npop == n: break
        cur = nxt
    return order, out.reshape(H, W), pushes

def make(seed, H=40, W=48, nseeds=12):
    rng = np.random.default_rng(seed)
    mask = rng.random((H, W)) < 0.75
    markers = np.zeros((H, W), np.int64)
    for k in range(1, nseeds + 1):
        y, x = rng.integers(0, H - 4), rng.integers(0, W - 4)
        h, w = rng.integers(1, 5), rng.integers(1, 5)
        markers[y:y + h, x:x + w] = k
    return mask, markers

if __name__ == "__main__":
    bad = 0
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
        mask, markers = make(seed, H=20 + seed % 30, W=16 + (seed * 7) % 40, nseeds=3 + seed % 20)
        ro, rl, rp = ref_order(mask, markers)
        n = int(((markers != 0) & mask).sum())
        mo, ml, mp = model_order(mask, markers)
        if ro[:n] != mo or rp[:len(mp)] != mp:
            bad += 1
            k = next((i for i, (a, b) in enumerate(zip(ro[:n], mo)) if a != b), None)
            print("seed", seed, "n", n, "first diff at", k, ro[:n][max(0,(k or 0)-3):(k or 0)+3], mo[max(0,(k or 0)-3):(k or 0)+3])
    print("bad", bad)

    # wider sweep: densities, sizes, degenerate cases
    import itertools
    bad = 0; tot = 0
    rng = np.random.default_rng(7)
    for trial in range(400):
        H, W = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        dens = rng.choice([0.0, 0.3, 0.6, 0.9, 1.0])
        mask = rng.random((H, W)) < dens if dens < 1 else np.ones((H, W), bool)
        markers = np.zeros((H, W), np.int64)
        style = trial % 4
        if style == 0:      # random pixels
            markers[rng.random((H, W)) < rng.choice([0.01, 0.1, 0.5, 1.0])] = 1
            markers *= rng.integers(1, 50, size=(H, W))
        elif style == 1:    # blobs
            for k in range(1, int(rng.integers(1, 30))):
                y, x = rng.integers(0, H), rng.integers(0, W)
                markers[y:y + rng.integers(1, 9), x:x + rng.integers(1, 9)] = k
        elif style == 2:    # stripes
            markers[::int(rng.integers(1, 5))] = 3
        else:               # one pixel / everything
            if trial % 8 == 3: markers[:] = 5
            else: markers[rng.integers(0, H), rng.integers(0, W)] = 1
        ro, rl, rp = ref_order(mask, markers)
        n = int(((markers != 0) & mask).sum())
        mo, ml, mp = model_order(mask, markers)
        tot += 1
        if ro[:n] != mo or rp[:len(mp)] != mp:
            bad += 1
            k = next((i for i, (a, b) in enumerate(zip(ro[:n], mo)) if a != b), None)
            print("trial", trial, H, W, dens, style, "n", n, "first diff", k)
    print("sweep bad", bad, "of", tot)
