"""Box tracker for ``VitInference(is_video=True)``: SORT (Bewley et al. 2016) -- a constant-velocity Kalman filter per track on
(centre x, centre y, area, aspect ratio) and a Hungarian assignment on the IoU between detections and predicted boxes.

CPU side, outside the HIP hot path.  Written from the algorithm, with the parameters and the calling
contract of the reference's tracker so that ``VitInference.inference`` sees the same thing
(``easy_ViTPose/sort.py:203-266``: ``update(dets[n,5]) -> [m,6] = (x1, y1, x2, y2, score, id)``, ids start at 1, a track is
reported when it was matched in this frame and has ``min_hits`` consecutive hits or the video is younger than ``min_hits``
frames, it is dropped after ``max_age`` frames without a match; constructed as ``Sort(max_age=yolo_step, min_hits=3 if yolo_step == 1 else 1,
iou_threshold=0.3)`` at ``easy_ViTPose/inference.py:179-184``).  The Kalman filter is the textbook predict / update pair with the
noise model of the SORT paper's public implementation (R = diag(1, 1, 10, 10), P0 = diag(10 x4, 1e4 x3),
Q = diag(1, 1, 1, 1, 0.01, 0.01, 1e-4)); no filterpy, no lap.

Two deliberate differences from the reference's tracker, neither visible through ``VitInference``'s contract within one video:
* track ids restart at 1 for every ``Sort`` instance (``VitInference.reset()`` builds a new one per video); the reference numbers
  tracks from the class-global ``KalmanBoxTracker.count`` (``sort.py:96,112-113``), which is never reset, so its ids keep growing
  across ``reset()`` calls and videos;
* a degenerate (zero-area) detection AND track pair has IoU 0 here; the reference divides 0 by 0 there and its assignment step then
  raises ``ValueError`` on the NaN.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment


def iou_matrix(dets: np.ndarray, trks: np.ndarray) -> np.ndarray:
    """IoU of every detection box with every track box, ``[len(dets), len(trks)]`` (boxes as x1, y1, x2, y2)."""
    d = dets[:, None, :4]
    t = trks[None, :, :4]
    w = np.clip(np.minimum(d[..., 2], t[..., 2]) - np.maximum(d[..., 0], t[..., 0]), 0.0, None)
    h = np.clip(np.minimum(d[..., 3], t[..., 3]) - np.maximum(d[..., 1], t[..., 1]), 0.0, None)
    inter = w * h
    union = (d[..., 2] - d[..., 0]) * (d[..., 3] - d[..., 1]) + (t[..., 2] - t[..., 0]) * (t[..., 3] - t[..., 1]) - inter
    return np.where(union > 0, inter / np.where(union > 0, union, 1.0), 0.0)   # zero-area pair: IoU 0, not 0 / 0


def box_to_z(box) -> np.ndarray:
    w, h = box[2] - box[0], box[3] - box[1]
    return np.array([box[0] + w / 2.0, box[1] + h / 2.0, w * h, w / float(h)], dtype=np.float64)


def x_to_box(x) -> np.ndarray:
    w = np.sqrt(x[2] * x[3])
    h = x[2] / w
    return np.array([x[0] - w / 2.0, x[1] - h / 2.0, x[0] + w / 2.0, x[1] + h / 2.0], dtype=np.float64)


_F = np.eye(7)
_F[0, 4] = _F[1, 5] = _F[2, 6] = 1.0                       # position += velocity
_H = np.eye(4, 7)
_R = np.diag([1.0, 1.0, 10.0, 10.0])
_Q = np.diag([1.0, 1.0, 1.0, 1.0, 0.01, 0.01, 1e-4])
_P0 = np.diag([10.0, 10.0, 10.0, 10.0, 1e4, 1e4, 1e4])


class Track:
    """One tracked box: Kalman state x = (cx, cy, area, ratio, vcx, vcy, varea)."""

    def __init__(self, box, score, track_id: int):
        self.x = np.zeros(7)
        self.x[:4] = box_to_z(box)
        self.P = _P0.copy()
        self.id = track_id
        self.score = score
        self.time_since_update = 0
        self.hit_streak = 0
        self.hits = 0
        self.age = 0

    def predict(self) -> np.ndarray:
        if self.x[6] + self.x[2] <= 0:                   # the area must not go negative
            self.x[6] = 0.0
        self.x = _F @ self.x
        self.P = _F @ self.P @ _F.T + _Q
        self.age += 1
        if self.time_since_update > 0:
            self.hit_streak = 0
        self.time_since_update += 1
        return x_to_box(self.x)

    def update(self, box, score):
        self.time_since_update = 0
        self.hits += 1
        self.hit_streak += 1
        y = box_to_z(box) - _H @ self.x
        S = _H @ self.P @ _H.T + _R
        K = self.P @ _H.T @ np.linalg.inv(S)
        self.x = self.x + K @ y
        IKH = np.eye(7) - K @ _H
        self.P = IKH @ self.P @ IKH.T + K @ _R @ K.T     # Joseph form (stays symmetric positive definite)
        self.score = score

    def box(self) -> np.ndarray:
        return x_to_box(self.x)


def associate(dets: np.ndarray, trks: np.ndarray, iou_threshold: float):
    """-> (matches [k,2] (det, trk), unmatched detection indices, unmatched track indices)"""
    if len(trks) == 0:
        return np.empty((0, 2), dtype=int), np.arange(len(dets)), np.empty((0,), dtype=int)
    if len(dets) == 0:
        return np.empty((0, 2), dtype=int), np.empty((0,), dtype=int), np.arange(len(trks))
    iou = iou_matrix(dets, trks)
    over = iou > iou_threshold
    if over.sum(1).max() <= 1 and over.sum(0).max() <= 1:   # unambiguous: no assignment problem to solve
        pairs = np.argwhere(over)
    else:
        r, c = linear_sum_assignment(-iou)
        pairs = np.stack([r, c], 1)
    pairs = np.array([p for p in pairs if iou[p[0], p[1]] >= iou_threshold], dtype=int).reshape(-1, 2)
    un_d = np.array([d for d in range(len(dets)) if d not in pairs[:, 0]], dtype=int)
    un_t = np.array([t for t in range(len(trks)) if t not in pairs[:, 1]], dtype=int)
    return pairs, un_d, un_t


class Sort:
    def __init__(self, max_age: int = 1, min_hits: int = 3, iou_threshold: float = 0.3):
        self.max_age, self.min_hits, self.iou_threshold = max_age, min_hits, iou_threshold
        self.tracks: "list[Track]" = []
        self.frame_count = 0
        self._next_id = 0

    def update(self, dets: np.ndarray = np.empty((0, 5))) -> np.ndarray:
        """Call once per frame (``np.empty((0, 5))`` when the detector did not run).  Returns ``[m, 6]``:
        box, score and id (>= 1) of the tracks to report; with empty detections every live track's PREDICTED box is returned
        (that is how the reference skips the detector on ``yolo_step`` frames, inference.py:235-248)."""
        dets = np.asarray(dets, dtype=np.float64).reshape(-1, 5)
        self.frame_count += 1
        pred = np.array([t.predict() for t in self.tracks]).reshape(-1, 4)
        bad = ~np.isfinite(pred).all(1) if len(pred) else np.zeros(0, bool)
        self.tracks = [t for t, b in zip(self.tracks, bad) if not b]
        pred = pred[~bad] if len(pred) else pred
        pairs, un_d, _ = associate(dets, pred, self.iou_threshold)
        for d, t in pairs:
            self.tracks[t].update(dets[d, :4], dets[d, 4])
        for d in un_d:
            self.tracks.append(Track(dets[d, :4], dets[d, 4], self._next_id))
            self._next_id += 1
        out, coasting = [], []
        for t in reversed(self.tracks):
            row = np.concatenate([t.box(), [t.score, t.id + 1]])
            if t.time_since_update < 1 and (t.hit_streak >= self.min_hits or self.frame_count <= self.min_hits):
                out.append(row)
            if len(dets) == 0:
                coasting.append(row)
        self.tracks = [t for t in self.tracks if t.time_since_update <= self.max_age]
        if out:
            return np.stack(out)
        if len(dets) == 0 and coasting:
            return np.stack(coasting)
        return np.empty((0, 6))
