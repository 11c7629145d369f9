from microbeseg_amd.utils.hull_polygon import get_indices_pandas, cv2_countour, label_polygons, points_string  # noqa: F401
