"""MapInitializer -- same public surface and return conventions as the reference's src/orbslam2/initializer.py:12-181.

initialize() follows the reference step by step (initializer.py:62-152): match -> E (RANSAC, threshold 3.0) ->
recover pose -> P1, P2 -> keep pose-mask matches -> triangulate -> cheirality filter -> colours -> map-point dicts.
Every cv2 call in that sequence runs on the MI355X through orbslam2.utils; the list/dict bookkeeping stays Python."""
import numpy as np

from .utils import (triangulate_points, convert_to_3d_points, calculate_essential_matrix, recover_pose,
                    compute_projection_matrix)


class MapInitializer:
    def __init__(self, camera_matrix, min_matches=10, min_inliers_ratio=0.9):
        self.camera_matrix = camera_matrix
        self.min_matches = min_matches
        self.min_inliers_ratio = min_inliers_ratio  # stored and never read, as in the reference (initializer.py:28)
        self.initialization_done = False
        self.first_frame_keypoints = None
        self.first_frame_descriptors = None
        self.first_frame_image = None
        self.current_frame_keypoints = None

    def set_first_frame(self, keypoints, descriptors, image):
        self.first_frame_keypoints = keypoints
        self.first_frame_descriptors = descriptors
        self.first_frame_image = image.copy()
        self.initialization_done = False

    def initialize(self, current_keypoints, current_descriptors, matcher, current_image):
        if self.first_frame_keypoints is None or self.first_frame_descriptors is None:
            return False, None, None, None, None
        matches = matcher.match(self.first_frame_descriptors, current_descriptors)
        if len(matches) < self.min_matches:
            print(f"Not enough matches for initialization: {len(matches)} < {self.min_matches}")
            return False, None, None, None, matches
        points1 = np.float32([self.first_frame_keypoints[m.queryIdx].pt for m in matches])
        points2 = np.float32([current_keypoints[m.trainIdx].pt for m in matches])
        E, mask = calculate_essential_matrix(points1, points2, self.camera_matrix, threshold=3.0)
        if E is None:
            return False, None, None, None, matches
        _, R, t, mask_pose, *_ = recover_pose(E, points1, points2, self.camera_matrix, mask)
        t = t.reshape(3, 1) if t.ndim == 1 else t
        P1 = compute_projection_matrix(np.eye(3), np.zeros((3, 1)), self.camera_matrix)
        P2 = compute_projection_matrix(R, t, self.camera_matrix)
        valid_matches = [m for m, ok in zip(matches, mask_pose.ravel().astype(bool)) if ok]
        if not valid_matches:  # the reference would raise on its debug prints here (initializer.py:100)
            return False, None, None, None, valid_matches
        points1 = np.float32([self.first_frame_keypoints[m.queryIdx].pt for m in valid_matches])
        points2 = np.float32([current_keypoints[m.trainIdx].pt for m in valid_matches])
        points_4d = triangulate_points(points1, points2, P1, P2)
        points_3d = convert_to_3d_points(points_4d)
        # points in front of both cameras (initializer.py:105-120)
        depth2 = (points_3d.astype(np.float64) @ R.T + t.ravel())[:, 2]
        valid_indices = [i for i in range(len(points_3d)) if points_3d[i, 2] > 0 and depth2[i] > 0]
        points_3d = points_3d[valid_indices]
        valid_matches = [valid_matches[i] for i in valid_indices]
        if len(points_3d) < self.min_matches // 2:
            print(f"Insufficient valid 3D points after filtering: {len(points_3d)}")
            return False, None, None, None, valid_matches
        img = self.first_frame_image
        colors = []
        for m in valid_matches:
            kp = self.first_frame_keypoints[m.queryIdx]
            x, y = int(kp.pt[0]), int(kp.pt[1])
            if 0 <= x < img.shape[1] and 0 <= y < img.shape[0]:
                colors.append(img[y, x, :] if img.ndim == 3 else np.array([img[y, x]] * 3))
            else:
                colors.append(np.array([0, 0, 255]))
        initial_map_points = [{'position': pt, 'color': colors[i],
                               'keypoint_references': {0: m.queryIdx, 1: m.trainIdx},
                               'observed_frames': [0, 1]}
                              for i, (pt, m) in enumerate(zip(points_3d, valid_matches))]
        self.initialization_done = True
        self.current_frame_keypoints = current_keypoints
        return True, R, t, initial_map_points, valid_matches

    def draw_initialization(self, first_image, current_image, matches):
        from .types import HAVE_CV2
        if not HAVE_CV2:
            raise RuntimeError("draw_initialization is visualisation and needs cv2 (out of scope of the HIP path)")
        import cv2
        a = cv2.cvtColor(first_image, cv2.COLOR_GRAY2BGR) if first_image.ndim == 2 else first_image.copy()
        b = cv2.cvtColor(current_image, cv2.COLOR_GRAY2BGR) if current_image.ndim == 2 else current_image.copy()
        return cv2.drawMatches(a, self.first_frame_keypoints, b, self.current_frame_keypoints, matches, None,
                               flags=cv2.DrawMatchesFlags_NOT_DRAW_SINGLE_POINTS, matchColor=(0, 255, 0),
                               singlePointColor=(255, 0, 0))
