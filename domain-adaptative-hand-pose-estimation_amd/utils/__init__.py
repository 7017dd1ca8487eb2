"""Mirror of the reference's ``utils`` package for the hot path (gl, keypoint_detection, logger, meter, data)."""
