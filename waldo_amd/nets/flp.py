"""Host-side mirror of the one step of the reference's models/nets/flp.py that feeds the path: the
pose heads' affine (flp.py:259-273; the same arithmetic sits in the LVD pose estimator,
models/nets/lvd.py:440-449).  The transformer that predicts the poses is outside this path; what
it hands over -- (B', No, 6 + 2 Lo) object poses and (B', 1, 6 + 2 L) background poses, after
``tanh`` and the optional ``+ last`` -- becomes the TPS control points ``Warper.forward`` consumes,
in one kernel forward and one backward (csrc/producers.hip) instead of seven framework launches.
"""
from .. import functional as WF


def obj_pose_to_points(pred_obj_pose, tgt_pts_obj, mul_obj, bias_obj, mul_delta_obj=1.0):
    """flp.py:260-265.  pred_obj_pose (B', No, 6 + 2 Lo); tgt_pts_obj (1, 1, Lo, 2); mul_obj /
    bias_obj (1, 1, 6) buffers (bias_obj may be the scalar 0 of the `no_bias` option, flp.py:208).
    Returns the object control points (B', No, Lo, 2)."""
    bias = bias_obj if hasattr(bias_obj, "reshape") else mul_obj.new_full((6,), float(bias_obj))
    return WF.pose_affine(pred_obj_pose, mul_obj, bias, tgt_pts_obj, mul_delta=mul_delta_obj, pts_mul=1.0)


def bg_pose_to_points(pred_bg_pose, tgt_pts_bg, bias_bg, bg_mul=1.0):
    """flp.py:268-273.  pred_bg_pose (B', 1, 6 + 2 L); tgt_pts_bg (1, 1, L, 2); bias_bg (1, 1, 6) or
    the scalar 0.  Returns the background control points (B', 1, L, 2)."""
    ones = tgt_pts_bg.new_ones(6)
    bias = bias_bg if hasattr(bias_bg, "reshape") else ones * float(bias_bg)
    return WF.pose_affine(pred_bg_pose, ones, bias, tgt_pts_bg, mul_delta=1.0, pts_mul=bg_mul)
