"""A SECOND, independent statement of the SMPL-X forward pass, written from the papers and not from oracle/geometry.py.

Why it exists: everything downstream of the body model (guidance gradients, trajectory re-derivation, metrics, the dataset-side
per-frame call) is checked against `oracle/geometry.py::BodyModel`, a restatement of the
third-party `smplx==0.1.28` (`/root/reference/environment.yml:198`; call site `data_loaders/motion_representation.py:379-396`) that
cannot be pinned to the real package in this environment.  One restatement checked only against itself is a single point of failure;
this file is a from-the-paper float64 numpy implementation that shares NO code with it -- homogeneous 4 x 4 transforms, explicit
per-joint loops -- so that a mistake would have to be made twice, independently, to pass.  It must not import `oracle`.

The model (SMPL: Loper et al., "SMPL: A Skinned Multi-Person Linear Model", SIGGRAPH Asia 2015, eqs. 2-10; SMPL-X: Pavlakos et al.,
"Expressive Body Capture", CVPR 2019, sec. 3 -- same function with 55 joints and expression blend shapes appended to the shape basis):

    T_P(beta, theta) = T_bar + B_S(beta) + B_P(theta)                       template + shape blend shapes + pose blend shapes
    B_S(beta)   = sum_n beta_n S_n                                          S: [V, 3, n_beta]
    J(beta)     = Jreg (T_bar + B_S(beta))                                  joints are regressed from the SHAPED template, [J, 3]
    B_P(theta)  = sum_n (R_n(theta) - R_n(0)) P_n                           n over the 9 (J - 1) entries of the non-root rotations, row-major
    G_k(theta, J) = prod_{j in ancestors(k), root first} [ R_j | J_j - J_parent(j) ]       world transform of joint k (rest offsets)
    G'_k        = G_k [ I | -J_k ]                                          ... relative to the rest pose
    v'_i        = ( sum_k w_{k,i} G'_k ) [ T_P,i ; 1 ] + transl             linear blend skinning
    joints_k    = translation of G_k + transl

R_j = exp(theta_j) by Rodrigues' formula.  Tensor conventions are the released model files' (as `smplx` loads them): `posedirs`
[9 (J - 1), 3 V] with the vertex coordinate fastest, `shapedirs` [V, 3, n], `lbs_weights` [V, J], `parents[0] = -1`.
"""
import numpy as np


def rodrigues(aa):
    """Axis-angle [3] -> rotation matrix [3, 3]: R = I + sin(a) K + (1 - cos(a)) K^2, K the cross-product matrix of the unit axis."""
    aa = np.asarray(aa, np.float64)
    angle = float(np.sqrt((aa * aa).sum()))
    if angle < 1e-300:
        return np.eye(3)
    x, y, z = aa / angle
    K = np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])
    return np.eye(3) + np.sin(angle) * K + (1.0 - np.cos(angle)) * (K @ K)


def rigid(R, t):
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = t
    return M


def forward_one(model, pose_aa, betas, transl):
    """One frame.  model: dict of float64 arrays (v_template [V,3], shapedirs [V,3,n], posedirs [9(J-1), 3V], J_regressor [J,V],
    lbs_weights [V,J], parents [J]); pose_aa [J,3] axis-angle of every joint (root first), betas [n] (shape, then expression),
    transl [3] -> (vertices [V,3], joints [J,3])."""
    vt, S, P = model['v_template'], model['shapedirs'], model['posedirs']
    Jreg, W, parents = model['J_regressor'], model['lbs_weights'], model['parents']
    V, J = vt.shape[0], Jreg.shape[0]
    # shape blend shapes, then the joints of the shaped template
    v_shaped = vt.copy()
    for n in range(len(betas)):
        v_shaped += betas[n] * S[:, :, n]
    rest = Jreg @ v_shaped                                                   # [J, 3]
    # pose blend shapes: (R_j - I) of the non-root joints, rotations flattened row-major, joint after joint
    R = [rodrigues(pose_aa[j]) for j in range(J)]
    feature = np.concatenate([(R[j] - np.eye(3)).reshape(9) for j in range(1, J)])      # [9 (J - 1)]
    v_posed = v_shaped + (feature @ P).reshape(V, 3)
    # kinematic chain: world transform of every joint from its parent's
    G = [None] * J
    for k in range(J):
        if parents[k] < 0:
            G[k] = rigid(R[k], rest[k])
        else:
            assert parents[k] < k, 'parents must precede their children'
            G[k] = G[parents[k]] @ rigid(R[k], rest[k] - rest[parents[k]])
    joints = np.stack([G[k][:3, 3] for k in range(J)]) + transl
    # relative to the rest pose, blended per vertex
    Grel = np.stack([G[k] @ rigid(np.eye(3), -rest[k]) for k in range(J)])   # [J, 4, 4]
    T = np.einsum('vk,kab->vab', W, Grel)                                    # [V, 4, 4]
    hom = np.concatenate([v_posed, np.ones((V, 1))], axis=1)
    verts = np.einsum('vab,vb->va', T, hom)[:, :3] + transl
    return verts, joints


def forward(model, pose_aa, betas, transl):
    """Frames [N, ...] -> (vertices [N, V, 3], joints [N, J, 3]), float64."""
    out = [forward_one(model, pose_aa[i], betas[i], transl[i]) for i in range(len(pose_aa))]
    return np.stack([o[0] for o in out]), np.stack([o[1] for o in out])


def as_model(tensors):
    """torch / numpy model tensors -> the float64 dict `forward` takes."""
    f = lambda a: np.asarray(a.detach().cpu().numpy() if hasattr(a, 'detach') else a, dtype=np.float64)
    m = {k: f(tensors[k]) for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights')}
    m['parents'] = np.asarray(tensors['parents'].cpu().numpy() if hasattr(tensors['parents'], 'cpu') else tensors['parents']).astype(np.int64)
    return m


def toy_model(seed=0, V=400, one_hot=True, with_blendshapes=False):
    """A small hand-checkable model on the 55-joint SMPL-X tree order given by `parents`: random rest joints, every vertex bound to
    exactly one joint (`one_hot`) with the regressor reproducing the rest joints exactly (each joint regressed from one vertex
    pinned ON it), blend shapes zero unless asked for.  With it the forward pass has closed forms (see the tests)."""
    from rohm_amd.utils.synth import SMPLX_PARENTS           # the kinematic tree (data, not arithmetic)
    parents = np.asarray(SMPLX_PARENTS, np.int64)
    J = len(parents)
    g = np.random.Generator(np.random.PCG64(seed))
    rest = np.zeros((J, 3))
    rest[0] = (0.1, -0.2, 0.9)
    for j in range(1, J):
        d = g.standard_normal(3)
        rest[j] = rest[parents[j]] + 0.15 * d / np.linalg.norm(d)
    owner = np.concatenate([np.arange(J), g.integers(0, J, size=V - J)])
    vt = rest[owner] + 0.05 * g.standard_normal((V, 3))
    vt[:J] = rest                                             # vertex j sits on joint j ...
    Jreg = np.zeros((J, V))
    Jreg[np.arange(J), np.arange(J)] = 1.0                    # ... and the regressor reads it
    W = np.zeros((V, J))
    if one_hot:
        W[np.arange(V), owner] = 1.0
    else:
        W = g.uniform(size=(V, J)) ** 6
        W /= W.sum(1, keepdims=True)
    n = 20
    S = np.zeros((V, 3, n))
    P = np.zeros((9 * (J - 1), 3 * V))
    if with_blendshapes:
        S = 0.01 * g.standard_normal((V, 3, n))
        S[:J] = 0.0                                           # keep the regressed joints where they are
        P = 0.002 * g.standard_normal((9 * (J - 1), 3 * V))
    return {'v_template': vt, 'shapedirs': S, 'posedirs': P, 'J_regressor': Jreg, 'lbs_weights': W, 'parents': parents}, owner, rest


def descendants(parents, k):
    """Joint k and everything below it."""
    out = {k}
    for j in range(len(parents)):
        if parents[j] in out:
            out.add(j)
    return out
