"""An ``open3d``-shaped namespace for the one Open3D call on the registration path.

``scripts/test_kitti.py:159-177`` builds point clouds and features with ``util/pointcloud.py:9-21``
(``o3d.geometry.PointCloud()``, ``o3d.utility.Vector3dVector``, ``o3d.pipelines.registration.Feature()`` with
``resize(dim, n)`` and a ``data`` array of shape ``[dim, n]`` in float64) and calls
``o3d.pipelines.registration.registration_ransac_based_on_feature_matching(pcd0, pcd1, feat0, feat1, False, d,
TransformationEstimationPointToPoint(False), 4, [CorrespondenceCheckerBasedOnEdgeLength(0.9),
CorrespondenceCheckerBasedOnDistance(d)], RANSACConvergenceCriteria(4000000, 10000))``.  With

    import eyoc_amd.o3d as o3d

those lines run unchanged; the work happens in ``libeyoc_hip.so`` (``eyoc_amd.registration``).  Only what that call site
touches exists here - this is not an Open3D re-implementation: other estimation methods, ``ransac_n != 4`` and
``mutual_filter=True`` raise ``NotImplementedError``, unknown checkers raise ``TypeError``.
"""
from __future__ import annotations

import types

import numpy as np

from . import registration as _reg


def _vector3d(a):
    """``o3d.utility.Vector3dVector``: an ``[n, 3]`` float64 array."""
    a = np.ascontiguousarray(np.asarray(a), dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError(f"Vector3dVector needs an [n, 3] array, got {a.shape}")
    return a


class PointCloud:
    """``o3d.geometry.PointCloud``: ``points`` / ``colors`` holders (util/pointcloud.py:9-14)."""

    def __init__(self, points=None):
        self.points = np.zeros((0, 3)) if points is None else _vector3d(points)
        self.colors = np.zeros((0, 3))

    def __len__(self):
        return len(self.points)


class Feature:
    """``o3d.pipelines.registration.Feature``: ``data`` is ``[dim, n]`` float64 (util/pointcloud.py:17-21)."""

    def __init__(self):
        self.data = np.zeros((0, 0))

    def resize(self, dim, n):
        self.data = np.zeros((int(dim), int(n)))

    def dimension(self):
        return self.data.shape[0]

    def num(self):
        return self.data.shape[1]


class TransformationEstimationPointToPoint:
    def __init__(self, with_scaling=False):
        self.with_scaling = bool(with_scaling)


class CorrespondenceCheckerBasedOnEdgeLength:
    def __init__(self, similarity_threshold=0.9):
        self.similarity_threshold = float(similarity_threshold)


class CorrespondenceCheckerBasedOnDistance:
    def __init__(self, distance_threshold):
        self.distance_threshold = float(distance_threshold)


class RANSACConvergenceCriteria:
    """``confidence`` is clamped to [0, 1] like Open3D >= 0.12 does; at 1 (the reference passes 10000) there is no early
    exit and every one of ``max_iteration`` hypotheses is evaluated - which is the only mode the library implements."""

    def __init__(self, max_iteration=100000, confidence=0.999):
        self.max_iteration = int(max_iteration)
        self.confidence = float(min(max(confidence, 0.0), 1.0))


RegistrationResult = _reg.RegistrationResult


def registration_ransac_based_on_feature_matching(source, target, source_feature, target_feature, mutual_filter,
                                                  max_correspondence_distance, estimation_method=None, ransac_n=3, checkers=(),
                                                  criteria=None, seed=0):
    """Positional layout of Open3D >= 0.12 (the 5th argument is ``mutual_filter``).  ``seed`` is an extension: Open3D draws
    from ``std::random_device``; here hypothesis ``h`` samples with the counter hash of ``oracle/ransac.py``."""
    if not isinstance(source, PointCloud) or not isinstance(target, PointCloud):
        raise TypeError("source / target must be eyoc_amd.o3d.geometry.PointCloud")
    if not isinstance(source_feature, Feature) or not isinstance(target_feature, Feature):
        raise TypeError("source_feature / target_feature must be eyoc_amd.o3d.pipelines.registration.Feature")
    if estimation_method is None:
        estimation_method = TransformationEstimationPointToPoint(False)
    if not isinstance(estimation_method, TransformationEstimationPointToPoint) or estimation_method.with_scaling:
        raise NotImplementedError("only TransformationEstimationPointToPoint(False) is implemented (scripts/test_kitti.py:173)")
    criteria = RANSACConvergenceCriteria() if criteria is None else criteria
    if criteria.confidence < 1.0:
        raise NotImplementedError("early termination (confidence < 1) is not implemented: the reference runs all iterations")
    edge, dist = None, None
    for c in checkers or ():
        if isinstance(c, CorrespondenceCheckerBasedOnEdgeLength):
            edge = c.similarity_threshold
        elif isinstance(c, CorrespondenceCheckerBasedOnDistance):
            dist = c.distance_threshold
        else:
            raise TypeError(f"unsupported correspondence checker {type(c).__name__}")
    if edge is None or dist is None:
        raise NotImplementedError("the kernels apply both the edge-length and the distance checker (scripts/test_kitti.py:174-175)")
    if abs(dist - float(max_correspondence_distance)) > 1e-12 * max(1.0, abs(dist)):
        raise NotImplementedError("the distance checker's threshold must equal max_correspondence_distance (as in the reference)")
    if source_feature.num() != len(source) or target_feature.num() != len(target):
        raise ValueError("features and points disagree in length")
    return _reg.registration_ransac_based_on_feature_matching(
        np.asarray(source.points, np.float32), np.asarray(target.points, np.float32),
        np.ascontiguousarray(source_feature.data.T, np.float32), np.ascontiguousarray(target_feature.data.T, np.float32),
        mutual_filter, float(max_correspondence_distance), None, ransac_n, None, (criteria.max_iteration, criteria.confidence),
        seed=seed, edge_similarity=edge)


geometry = types.SimpleNamespace(PointCloud=PointCloud)
utility = types.SimpleNamespace(Vector3dVector=_vector3d)
pipelines = types.SimpleNamespace(registration=types.SimpleNamespace(
    Feature=Feature, TransformationEstimationPointToPoint=TransformationEstimationPointToPoint,
    CorrespondenceCheckerBasedOnEdgeLength=CorrespondenceCheckerBasedOnEdgeLength,
    CorrespondenceCheckerBasedOnDistance=CorrespondenceCheckerBasedOnDistance,
    RANSACConvergenceCriteria=RANSACConvergenceCriteria, RegistrationResult=RegistrationResult,
    registration_ransac_based_on_feature_matching=registration_ransac_based_on_feature_matching))
