"""Unseen target cameras for the universal attack (`--use_unseen_views`, eval/ibrnet/eval_adv.py:652-691, eval/gnt/eval_adv.py:
782-822): every iteration draws three of the scene's render poses and two interpolation parameters from numpy's GLOBAL
generator (same draw order as the reference, so a seeded run visits the same cameras) and interpolates them pairwise --
rotation by spherical interpolation of the quaternions, camera centre linearly (eval/ibrnet/geo_interp.py:6-45; the decoupled
form with separate rotation / translation parameters :17-23 exists only in the IBRNet flavour of the reference, here in both).
Host-side float64 arithmetic on 4x4 matrices; nothing here touches the GPU."""
import numpy as np
import torch
from scipy.spatial.transform import Rotation


def slerp(q0, q1, t):
    """spherical interpolation of two quaternions as the reference writes it (geo_interp.py:6-10): the angle from the
    NORMALISED operands, the blend of the operands as given; no shortest-arc sign flip"""
    q0, q1 = np.asarray(q0, dtype=np.float64), np.asarray(q1, dtype=np.float64)
    omega = np.arccos(np.dot(q0 / np.linalg.norm(q0), q1 / np.linalg.norm(q1)))
    so = np.sin(omega)
    return np.sin((1.0 - t) * omega) / so * q0 + np.sin(t * omega) / so * q1


def interp(pose1, pose2, s):
    """camera-to-world [4,4] between pose1 (s = 0) and pose2 (s = 1); s: scalar, or [s_rot, s_trans] -> float64 tensor [4,4]"""
    pose1, pose2 = np.asarray(pose1, dtype=np.float64), np.asarray(pose2, dtype=np.float64)
    assert pose1.shape == (4, 4) and pose2.shape == (4, 4)
    s_rot, s_trans = (s[0], s[1]) if isinstance(s, (list, tuple)) else (s, s)
    out = np.eye(4)
    out[:3, 3] = (1 - s_trans) * pose1[:3, 3] + s_trans * pose2[:3, 3]
    q = slerp(Rotation.from_matrix(pose1[:3, :3]).as_quat(), Rotation.from_matrix(pose2[:3, :3]).as_quat(), s_rot)
    out[:3, :3] = Rotation.from_quat(q).as_matrix()
    return torch.from_numpy(out)


def interp3(pose1, pose2, pose3, s12, s3):
    """geo_interp.py:44-45"""
    return interp(interp(pose1, pose2, s12).numpy(), pose3, s3)


def sample_unseen_pose(args, render_poses, rng=np.random):
    """one iteration's draws (eval_adv.py:655-683) -> float64 tensor [4,4].  rng: numpy's global generator unless given."""
    n = len(render_poses)
    if getattr(args, 'sample_based_on_depth', False):
        z_camera = np.array([np.asarray(p)[2, 2] for p in render_poses])          # forward direction
        p_camera = np.exp(z_camera / args.temp) / np.sum(np.exp(z_camera / args.temp))
        ids = rng.choice(n, size=3, p=p_camera, replace=False)
    else:
        ids = rng.choice(n, size=3, replace=False)
    if getattr(args, 'decouple_interp_range', False):
        s12_rot, s3_rot = rng.uniform(0, args.interp_upbound_rot, size=2)
        s12_trans, s3_trans = rng.uniform(0, args.interp_upbound_trans, size=2)
        s12, s3 = [s12_rot, s12_trans], [s3_rot, s3_trans]
    elif getattr(args, 'sample_based_on_depth', False):
        s12, s3 = rng.beta(args.beta, args.beta, size=2) * args.interp_upbound_rot
    else:
        s12, s3 = rng.uniform(0, getattr(args, 'interp_upbound', 1.0), size=2)
    to_np = lambda p: p.detach().cpu().numpy() if torch.is_tensor(p) else np.asarray(p)
    return interp3(to_np(render_poses[ids[0]]), to_np(render_poses[ids[1]]), to_np(render_poses[ids[2]]), s12, s3)


def unseen_camera(args, render_poses, camera, rng=np.random):
    """camera [1,34] of a loader batch -> the same image size and intrinsics with a freshly drawn interpolated pose
    (eval_adv.py:685-691)"""
    pose = sample_unseen_pose(args, render_poses, rng).flatten().unsqueeze(0).to(camera)
    return torch.cat([camera[:, :18], pose], dim=1)
