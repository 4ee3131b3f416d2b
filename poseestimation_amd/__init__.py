"""poseestimation_amd -- MI355X-native SVD -> SO(3) projection head (drop-in for the hot path of
henrikgruner/PoseEstimation's rotation_representation.py).  See DESIGN.md / INTEGRATION.md."""
from .rotation_representation import (  # noqa: F401
    angle_error,
    angle_error_statistics,
    calculate_T_pred,
    compute_geodesic_distance_from_two_matrices,
    compute_rotation_matrix_from_ortho6d,
    frobenius_head,
    get_sampled_rotation_matrices_by_axisAngle,
    head_angle_error,
    kabsch_rotation,
    kabsch_rotation_synthetic,
    loss_frobenius,
    symmetric_orthogonalization,
    transform_output,
)

__all__ = [
    "symmetric_orthogonalization",
    "angle_error",
    "angle_error_statistics",
    "calculate_T_pred",
    "compute_geodesic_distance_from_two_matrices",
    "compute_rotation_matrix_from_ortho6d",
    "loss_frobenius",
    "frobenius_head",
    "head_angle_error",
    "kabsch_rotation",
    "kabsch_rotation_synthetic",
    "get_sampled_rotation_matrices_by_axisAngle",
    "transform_output",
]
