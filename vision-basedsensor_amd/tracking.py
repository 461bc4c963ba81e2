"""Drop-in for the reference's `code/Marker_Tracking/tracking.py`: `process_video()` with the same
hard-coded paths and constants (`tracking.py:29-45`), built on `MarkerTracker` (the published script
inlines the same loop and imports `find_marker` / `marker_center`, which this package does export)."""
import os

from .marker_detection import MarkerTracker, find_marker, marker_center  # noqa: F401  (`tracking.py:7`)


def process_video(video_dir="./video", input_name="test2.avi", output_csv="marker_locations_0.csv",
                  crop_ratios=(1 / 8, 1 / 8, 1 / 16, 0), min_marker_distance=20, num_outer_layers=5,
                  **extra):
    """Track the marker array of `<video_dir>/<input_name>` and write `<video_dir>/<output_csv>` with
    columns frameno,row,col,Ox,Oy,Cx,Cy,major_axis,minor_axis,angle (`tracking.py:74-85,255`).
    The defaults are the reference's constants; the annotated video is not produced."""
    os.makedirs(video_dir, exist_ok=True)
    path = os.path.join(video_dir, input_name)
    if not os.path.exists(path):
        raise FileNotFoundError(f"Could not open video: {path}")          # `tracking.py:52-53`
    cfg = {"video_path": path, "output_dir": video_dir, "crop_ratios": crop_ratios,
           "num_layers": num_outer_layers, "min_marker_distance": min_marker_distance, **extra}
    tracker = MarkerTracker(cfg)
    tracker.output_csv = os.path.join(video_dir, output_csv)
    tracker.process()
    return tracker.output_csv


if __name__ == "__main__":
    process_video()
