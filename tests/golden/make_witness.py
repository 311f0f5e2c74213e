#!/usr/bin/env python3
"""Independent float64 witness for the render stages x1-x5 (SH evaluation, 3D->2D projection,
tile keys, sort, blend).

WHY: the reference holds no implementation of these stages, so the CPU oracle
(oracle/gs_oracle.c) and the HIP kernels are both the build's own statement of DESIGN.md §3 and
were written side by side.  A shared convention error (SH basis sign, Jacobian, y flip, use of
S^-1 R^T for the view direction) would pass every oracle-vs-HIP test.  This file is a second,
structurally different restatement, written from the *text* of DESIGN.md §3 and from Kerbl et al.
2023 ("3D Gaussian Splatting for Real-Time Radiance Field Rendering", eqs. 5-6 and the reference
rasteriser's documented conventions) — NOT from gs_oracle.c:
  * float64 throughout, numpy-vectorised over Gaussians / pixels;
  * full 3x3 / 4x4 matrix algebra (np.einsum / @) instead of hand-expanded scalar terms;
  * covariances built from (rot, scale) with R diag(s^2) R^T, the view matrix rebuilt from
    (eye, target, up), the conic by np.linalg.inv, the largest eigenvalue by np.linalg.eigvalsh
    (the spec's closed form is checked against it), exp by np.exp;
  * SH basis from the textbook real spherical harmonics (Condon-Shortley phase folded into the
    constants exactly as the 3DGS paper's code does), written per degree as dot products.
It narrows common-mode risk; it does NOT pin the rows to the reference (they stay "parity
unpinned").

Output: tests/golden/witness_v2.npz — inputs (Gaussians, uniforms) and expected projected
records / images for three views, plus margins that tell the test which discontinuous decisions
(ceil of the radius, cull tests, tile-rect floors) are too close to call in float32.
Run:  python tests/golden/make_witness.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

GAUSSIAN_DTYPE = np.dtype([("rot", "<f4", 4), ("pos", "<f4", 3), ("color", "u1", 4),
                           ("sh", "<f4", 45), ("scale", "<f4", 3)])


# ------------------------------------------------------------------------------------------------
# conventions of the data model (pinned rows a1, a10-a12; restated here in f64)
# ------------------------------------------------------------------------------------------------

def quat_to_mat(q):
    """unit quaternion (x, y, z, w) -> rotation matrix (textbook form)"""
    x, y, z, w = [np.asarray(q[..., k], dtype=np.float64) for k in range(4)]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z)
    R[..., 0, 1] = 2 * (x * y - z * w)
    R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w)
    R[..., 1, 1] = 1 - 2 * (x * x + z * z)
    R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w)
    R[..., 2, 1] = 2 * (y * z + x * w)
    R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def look_at_rh(eye, target, up):
    """right-handed view matrix, camera looks down -Z, +Y up (glam Mat4::look_at_rh)"""
    eye, target, up = [np.asarray(v, dtype=np.float64) for v in (eye, target, up)]
    f = target - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    V = np.eye(4)
    V[0, :3], V[1, :3], V[2, :3] = s, u, -f
    V[:3, 3] = -V[:3, :3] @ eye
    return V


def max_std_dev_roundtrip(v):
    """GaussianTransformPod stores max_std_dev as u8 = trunc(v / 3 * 255) (f32 arithmetic) and the
    shader decodes u8 / 255 * 3 (src/buffer/gaussian_transform.rs:63-77, gaussian_transform.wesl)"""
    u8 = int(np.float32(np.float32(v) / np.float32(3.0)) * np.float32(255.0))
    return u8, float(u8) / 255.0 * 3.0


# real SH, degrees 1..3, in the ordering / sign convention of the 3DGS paper's rasteriser
# (m = -l..l per degree).  Returned as (n, 15) basis values for unit directions d (n, 3).
def sh_rest_basis(d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    B = np.zeros((len(d), 15))
    c1 = np.sqrt(3.0 / (4.0 * np.pi))
    B[:, 0], B[:, 1], B[:, 2] = -c1 * y, c1 * z, -c1 * x
    c2 = np.sqrt(15.0 / np.pi)
    B[:, 3] = 0.5 * c2 * x * y
    B[:, 4] = -0.5 * c2 * y * z
    B[:, 5] = 0.25 * np.sqrt(5.0 / np.pi) * (3.0 * z * z - 1.0)       # (2zz - xx - yy) on the unit sphere
    B[:, 6] = -0.5 * c2 * x * z
    B[:, 7] = 0.25 * c2 * (x * x - y * y)
    B[:, 8] = -0.25 * np.sqrt(35.0 / (2.0 * np.pi)) * y * (3.0 * x * x - y * y)
    B[:, 9] = 0.5 * np.sqrt(105.0 / np.pi) * x * y * z
    B[:, 10] = -0.25 * np.sqrt(21.0 / (2.0 * np.pi)) * y * (5.0 * z * z - 1.0)   # (4zz - xx - yy)
    B[:, 11] = 0.25 * np.sqrt(7.0 / np.pi) * z * (5.0 * z * z - 3.0)            # (2zz - 3xx - 3yy)
    B[:, 12] = -0.25 * np.sqrt(21.0 / (2.0 * np.pi)) * x * (5.0 * z * z - 1.0)
    B[:, 13] = 0.25 * np.sqrt(105.0 / np.pi) * z * (x * x - y * y)
    B[:, 14] = -0.25 * np.sqrt(35.0 / (2.0 * np.pi)) * x * (x * x - 3.0 * y * y)
    return B


# ------------------------------------------------------------------------------------------------
# the witness renderer
# ------------------------------------------------------------------------------------------------

def project(g, view):
    """x1 + x2 for every Gaussian of `g` under the uniforms `view` (a dict).  Everything f64."""
    n = len(g)
    W, H = view["width"], view["height"]
    fx, fy, cx, cy = view["fx"], view["fy"], view["cx"], view["cy"]
    pos = g["pos"].astype(np.float64)

    # model transform: T R S (model_transform.wesl); SR = R S; inverse of SR = S^-1 R^T
    Rm = quat_to_mat(np.asarray(view["model_rot"], dtype=np.float64))
    Sm = np.diag(np.asarray(view["model_scale"], dtype=np.float64))
    SR = Rm @ Sm
    ISR = np.linalg.inv(SR)
    pw = pos @ SR.T + np.asarray(view["model_pos"], dtype=np.float64)

    V = look_at_rh(view["eye"], view["target"], view["up"])
    pc = pw @ V[:3, :3].T + V[:3, 3]                 # camera space, -Z forward, +Y up
    flip = np.diag([1.0, -1.0, -1.0])                # -> +Y down (image rows), +Z forward (depth)
    pv = pc @ flip.T
    x, y, z = pv[:, 0], pv[:, 1], pv[:, 2]
    in_depth = (z > view["near"]) & (z < view["far"])
    zs = np.where(in_depth, z, 1.0)                   # keep the arithmetic finite for culled ones

    # 3D covariance in model space, then world, scaled by size^2
    Rg = quat_to_mat(g["rot"].astype(np.float64))
    s2 = g["scale"].astype(np.float64) ** 2
    Sigma = np.einsum("nij,nj,nkj->nik", Rg, s2, Rg)
    Sigma_w = view["size"] ** 2 * np.einsum("ij,njk,lk->nil", SR, Sigma, SR)

    # perspective Jacobian at the clamped position (EWA splatting; the 1.3 * tan(fov / 2) guard band)
    limx = 1.3 * (0.5 * W / fx)
    limy = 1.3 * (0.5 * H / fy)
    xc = np.clip(x / zs, -limx, limx) * zs
    yc = np.clip(y / zs, -limy, limy) * zs
    J = np.zeros((n, 2, 3))
    J[:, 0, 0] = fx / zs
    J[:, 0, 2] = -fx * xc / zs ** 2
    J[:, 1, 1] = fy / zs
    J[:, 1, 2] = -fy * yc / zs ** 2
    Wc = flip @ V[:3, :3]                             # world -> (x right, y down, z forward)
    T = J @ Wc
    cov2d = np.einsum("nij,njk,nlk->nil", T, Sigma_w, T) + 0.3 * np.eye(2)
    det = np.linalg.det(cov2d)
    ok_det = det > 0
    conic = np.linalg.inv(np.where(ok_det[:, None, None], cov2d, np.eye(2)))
    lam_true = np.linalg.eigvalsh(cov2d)[:, 1]
    mid = 0.5 * (cov2d[:, 0, 0] + cov2d[:, 1, 1])
    lam_spec = mid + np.sqrt(np.maximum(0.1, mid * mid - det))
    # the spec's closed form is the largest eigenvalue unless the two eigenvalues are within
    # 2 sqrt(0.1) of each other (the 0.1 floor of the reference rasteriser)
    sep = mid * mid - det >= 0.1
    assert np.allclose(lam_spec[sep & ok_det], lam_true[sep & ok_det], rtol=1e-9)
    k = view["max_std_dev_decoded"]
    rad_real = k * np.sqrt(lam_spec)
    radius = np.ceil(rad_real)

    mx = fx * (x / zs) + cx
    my = fy * (y / zs) + cy
    tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
    b0, b1 = view.get("band", (0, tiles_y))
    tx0 = np.clip(np.floor((mx - radius) / 16.0), 0, tiles_x)
    tx1 = np.clip(np.floor((mx + radius) / 16.0) + 1, 0, tiles_x)
    ty0 = np.clip(np.floor((my - radius) / 16.0), b0, b1)
    ty1 = np.clip(np.floor((my + radius) / 16.0) + 1, b0, b1)
    # DESIGN.md §3.3, second step of the rect (display mode Splat, which is all this witness renders):
    # alpha = (k / 255) exp(power) reaches 1/255 only where power >= -ln k, k the opacity byte.  The set
    # {power >= -(ln k + 0.1)} is the ellipse d^T cov2d^-1 d <= 2 (ln k + 0.1), whose bounding box has the
    # half-extents sqrt(2 (ln k + 0.1) cov2d_xx) and sqrt(2 (ln k + 0.1) cov2d_yy) — derived here from the
    # covariance, where the implementations work from the inverted (conic) form.  Tile t holds the pixel
    # centres 16 t + 0.5 ... 16 t + 15.5; tiles the box does not reach are dropped; k = 0 is never visible.
    kop = g["color"][:, 3].astype(np.int64)
    lim = np.log(np.maximum(kop, 1).astype(np.float64)) + 0.1
    ex = np.sqrt(2.0 * lim * cov2d[:, 0, 0])
    ey = np.sqrt(2.0 * lim * cov2d[:, 1, 1])
    cx0, cx1 = np.floor((mx - ex - 15.5) / 16.0) + 1, np.floor((mx + ex - 0.5) / 16.0) + 1
    cy0, cy1 = np.floor((my - ey - 15.5) / 16.0) + 1, np.floor((my + ey - 0.5) / 16.0) + 1
    tx0, tx1 = np.maximum(tx0, cx0), np.minimum(tx1, cx1)
    ty0, ty1 = np.maximum(ty0, cy0), np.minimum(ty1, cy1)
    visible = in_depth & ok_det & (radius > 0) & (tx1 > tx0) & (ty1 > ty0) & (kop > 0)

    # colour: view direction camera -> Gaussian, taken to model space with S^-1 R^T, renormalised
    dw = pw - np.asarray(view["eye"], dtype=np.float64)
    dw /= np.linalg.norm(dw, axis=1, keepdims=True)
    dm = dw @ ISR.T
    dm /= np.linalg.norm(dm, axis=1, keepdims=True)
    base = g["color"][:, :3].astype(np.float64) / 255.0
    rgb = np.zeros((n, 3)) if view["no_sh0"] else base.copy()
    ncoef = [0, 3, 8, 15][view["sh_deg"]]
    if ncoef:
        B = sh_rest_basis(dm)[:, :ncoef]
        coef = g["sh"].astype(np.float64).reshape(n, 15, 3)[:, :ncoef]
        rgb = rgb + np.einsum("nk,nkc->nc", B, coef)
    rgb = np.maximum(rgb, 0.0)
    opacity = g["color"][:, 3].astype(np.float64) / 255.0

    # how close each discontinuous decision is to flipping (the f32 implementations carry ~1e-6
    # relative error; the test skips exact comparisons whose margin is below its threshold)
    frac = rad_real - np.floor(rad_real)
    m_radius = np.minimum(frac, 1.0 - frac)
    edges = np.stack([(mx - radius) / 16.0, (mx + radius) / 16.0, (my - radius) / 16.0, (my + radius) / 16.0], 1)
    clip_edges = np.stack([(mx - ex - 15.5) / 16.0, (mx + ex - 0.5) / 16.0, (my - ey - 15.5) / 16.0,
                           (my + ey - 0.5) / 16.0], 1)
    m_rect = np.minimum(np.abs(edges - np.round(edges)).min(axis=1),
                        np.abs(clip_edges - np.round(clip_edges)).min(axis=1)) * 16.0
    m_depth = np.minimum(np.abs(z - view["near"]), np.abs(z - view["far"]))
    return dict(visible=visible, mx=mx, my=my, conic=conic, cov2d=cov2d, det=det, rgb=rgb, opacity=opacity,
                depth=z, radius=radius, rect=np.stack([tx0, ty0, tx1, ty1], 1).astype(np.int64),
                m_radius=m_radius, m_rect=m_rect, m_depth=m_depth)


def blend(p, view):
    """x3-x5: per 16x16 tile, the visible splats whose rect covers the tile, front to back."""
    W, H = view["width"], view["height"]
    bg = np.asarray(view["background"], dtype=np.float64)
    img = np.zeros((H, W, 4))
    tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
    vis = np.nonzero(p["visible"])[0]
    order = vis[np.argsort(p["depth"][vis], kind="stable")]      # global depth order, ties by index
    rect = p["rect"]
    min_alpha_margin = np.inf
    pairs = 0
    for ty in range(tiles_y):
        for tx in range(tiles_x):
            sel = order[(rect[order, 0] <= tx) & (tx < rect[order, 2]) & (rect[order, 1] <= ty) & (ty < rect[order, 3])]
            pairs += len(sel)
            ys, xs = np.mgrid[ty * 16:min(ty * 16 + 16, H), tx * 16:min(tx * 16 + 16, W)]
            px, py = xs + 0.5, ys + 0.5
            T = np.ones(px.shape)
            C = np.zeros(px.shape + (3,))
            done = np.zeros(px.shape, dtype=bool)
            for i in sel:
                dx, dy = p["mx"][i] - px, p["my"][i] - py
                A, Bc, Cc = p["conic"][i, 0, 0], p["conic"][i, 0, 1], p["conic"][i, 1, 1]
                power = -0.5 * (A * dx * dx + Cc * dy * dy) - Bc * dx * dy
                alpha = np.minimum(0.99, p["opacity"][i] * np.exp(np.minimum(power, 0.0)))
                use = (power <= 0.0) & (alpha >= 1.0 / 255.0) & ~done
                if use.any():
                    min_alpha_margin = min(min_alpha_margin, np.abs(alpha[~done] * 255.0 - 1.0).min())
                nT = T * (1.0 - alpha)
                stop = use & (nT < 1e-4)
                done |= stop
                use &= ~stop
                C[use] += (p["rgb"][i] * (alpha * T)[..., None])[use]
                T[use] = nT[use]
            img[ys, xs, :3] = C + T[..., None] * bg
            img[ys, xs, 3] = 1.0 - T
    return img, pairs


# ------------------------------------------------------------------------------------------------
# scenes and views
# ------------------------------------------------------------------------------------------------

def make_scene(rng, n):
    g = np.zeros(n, dtype=GAUSSIAN_DTYPE)
    g["pos"] = np.stack([rng.uniform(-9, 9, n), rng.uniform(-6, 6, n), -rng.uniform(1.5, 16, n)], 1)
    q = rng.normal(size=(n, 4))
    g["rot"] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    g["scale"] = np.exp(rng.normal(-2.3, 0.55, size=(n, 3)))
    g["color"][:, :3] = rng.integers(0, 256, size=(n, 3))
    g["color"][:, 3] = rng.integers(24, 256, size=n)
    g["sh"] = rng.uniform(-0.3, 0.3, size=(n, 45))
    return g


def focal(vfov_deg, height):
    return 0.5 * height / np.tan(0.5 * np.deg2rad(vfov_deg))


def views():
    W, H = 192, 112
    f = focal(60.0, H)
    base = dict(width=W, height=H, near=0.1, far=100.0, cx=0.5 * W, cy=0.5 * H, background=(0.0, 0.0, 0.0),
                model_pos=(0.0, 0.0, 0.0), model_rot=(0.0, 0.0, 0.0, 1.0), model_scale=(1.0, 1.0, 1.0),
                size=1.0, sh_deg=3, no_sh0=False, max_std_dev=3.0, up=(0.0, 1.0, 0.0))
    a = dict(base, name="A_identity", eye=(0.0, 0.0, 0.0), target=(0.0, 0.0, -1.0), fx=f, fy=f)
    qm = np.array([0.18, -0.31, 0.12, 0.92])
    qm /= np.linalg.norm(qm)
    b = dict(base, name="B_model_transform", eye=(2.5, 1.5, 3.0), target=(0.3, -0.4, -7.0), up=(0.15, 1.0, 0.1),
             fx=f, fy=f, model_pos=(0.6, -0.35, 0.25), model_rot=tuple(qm.astype(np.float32).astype(float)),
             model_scale=(1.25, 0.8, 1.1), size=1.3, sh_deg=2, max_std_dev=2.5, background=(0.2, 0.4, 0.6))
    fw = focal(100.0, H)
    c = dict(base, name="C_wide_no_sh0", eye=(-1.0, 0.5, -4.0), target=(1.5, 0.0, -9.0), fx=fw, fy=1.15 * fw,
             cx=0.5 * W + 3.25, cy=0.5 * H - 2.5, near=0.5, far=14.0, sh_deg=1, no_sh0=True,
             background=(1.0, 0.5, 0.25))
    out = []
    for v in (a, b, c):
        v["max_std_dev_u8"], v["max_std_dev_decoded"] = max_std_dev_roundtrip(v["max_std_dev"])
        out.append(v)
    return out


def main():
    rng = np.random.default_rng(20260117)
    g = make_scene(rng, 3000)
    out = dict(rot=g["rot"], pos=g["pos"], color=g["color"], sh=g["sh"], scale=g["scale"])
    for v in views():
        p = project(g, v)
        img, pairs = blend(p, v)
        n = v["name"]
        for key in ("eye", "target", "up", "model_pos", "model_rot", "model_scale", "background"):
            out["%s/%s" % (n, key)] = np.asarray(v[key], dtype=np.float64)
        for key in ("width", "height", "sh_deg", "max_std_dev_u8"):
            out["%s/%s" % (n, key)] = np.int64(v[key])
        out["%s/no_sh0" % n] = np.bool_(v["no_sh0"])
        for key in ("near", "far", "fx", "fy", "cx", "cy", "size", "max_std_dev", "max_std_dev_decoded"):
            out["%s/%s" % (n, key)] = np.float64(v[key])
        out["%s/view" % n] = look_at_rh(v["eye"], v["target"], v["up"])
        for key in ("visible", "mx", "my", "rgb", "opacity", "depth", "radius", "rect", "det", "m_radius", "m_rect",
                    "m_depth"):
            out["%s/%s" % (n, key)] = p[key]
        out["%s/conic" % n] = np.stack([p["conic"][:, 0, 0], p["conic"][:, 0, 1], p["conic"][:, 1, 1]], 1)
        out["%s/image" % n] = img.astype(np.float32)     # the comparison tolerance is 1e-4
        out["%s/pairs" % n] = np.int64(pairs)
        print(n, "visible", int(p["visible"].sum()), "pairs", pairs, "covered", int((img[..., 3] > 0).sum()),
              "alpha max", img[..., 3].max())
    out["views"] = np.array([v["name"] for v in views()])
    np.savez_compressed(os.path.join(HERE, "witness_v2.npz"), **out)


if __name__ == "__main__":
    main()
