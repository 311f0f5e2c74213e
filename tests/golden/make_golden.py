#!/usr/bin/env python3
"""Golden-vector generator (numpy twin) for rows a1-a12 of SURVEY.md §8.

The reference (Rust + WGSL) cannot be built or run in this image, so these vectors are produced
by an *independent numpy restatement of the formulas in the reference text* — never by running
the reference, and never by calling oracle/ or the product library.  Every block cites the
reference file:line it restates.  Output: tests/golden/golden_v1.npz (+ model.ply / model.spz,
which are the reference's own example data files, copied verbatim as data fixtures).

Run:  python tests/golden/make_golden.py
"""
import os
import struct

import numpy as np

F = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))
SEEDS = list(range(15)) + [42, 123]
SH_NAMES = ["single", "half", "norm8", "none"]
COV_NAMES = ["rot_scale", "single", "half"]
SH_BYTES = {"single": 180, "half": 92, "norm8": 48, "none": 0}       # gaussian_config.rs:37,54,90,127
COV_BYTES = {"rot_scale": 28, "single": 24, "half": 12}             # gaussian_config.rs:171,193,224
# padding_size (f32 units), src/buffer/gaussian.rs:373-384
PADDING = {("single", "rot_scale"): 0, ("single", "single"): 1, ("single", "half"): 0,
           ("half", "rot_scale"): 2, ("half", "single"): 3, ("half", "half"): 2,
           ("norm8", "rot_scale"): 1, ("norm8", "single"): 2, ("norm8", "half"): 1,
           ("none", "rot_scale"): 1, ("none", "single"): 2, ("none", "half"): 1}


def given_gaussian(seed):
    """tests/common/given.rs:48-81, in f32 arithmetic."""
    base = F(seed)
    q = np.array([base + F(0.1), base + F(0.2), base + F(0.3), base + F(0.4)], dtype=F)
    length = np.sqrt(F(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]), dtype=F)
    rot = (q / length).astype(F)
    pos = np.array([base + F(1.1), base + F(2.2), base + F(3.3)], dtype=F)
    color = np.array([np.fmod(base + F(v), F(256.0)) for v in (11.0, 22.0, 33.0, 44.0)],
                     dtype=F).astype(np.uint8)
    sh = np.zeros(45, dtype=F)
    for i in range(15):
        sh_base = F(base + F(F(i) * F(0.3)))
        for c, off in enumerate((0.1, 0.2, 0.3)):
            sh[3 * i + c] = F(np.fmod(F(sh_base + F(off)), F(2.0))) - F(1.0)
    scale = np.array([base + F(0.12), base + F(0.34), base + F(0.56)], dtype=F)
    return dict(rot=rot, pos=pos, color=color, sh=sh, scale=scale)


def fixed_gaussian():
    """src/buffer/gaussian.rs:396-402 (the unit-test Gaussian)."""
    return dict(rot=np.array([0, 0, 0, 1], dtype=F), pos=np.array([1, 2, 3], dtype=F),
                color=np.array([255, 128, 64, 32], dtype=np.uint8),
                sh=np.tile(np.array([0.1, 0.2, 0.3], dtype=F), 15),
                scale=np.array([1, 2, 3], dtype=F))


def quat_to_cols(q):
    """glam Mat3::from_quat == the x2/xx/wz formulation of gaussian.wesl:84-95 (f32)."""
    x, y, z, w = (F(v) for v in q)
    x2, y2, z2 = F(x + x), F(y + y), F(z + z)
    xx, xy, xz = F(x * x2), F(x * y2), F(x * z2)
    yy, yz, zz = F(y * y2), F(y * z2), F(z * z2)
    wx, wy, wz = F(w * x2), F(w * y2), F(w * z2)
    one = F(1.0)
    c0 = np.array([one - F(yy + zz), F(xy + wz), F(xz - wy)], dtype=F)
    c1 = np.array([F(xy - wz), one - F(xx + zz), F(yz + wx)], dtype=F)
    c2 = np.array([F(xz + wy), F(yz - wx), one - F(xx + yy)], dtype=F)
    return c0, c1, c2


def cov6_f32(rot, scale):
    """gaussian_config.rs:195-208 / gaussian.wesl:80-129: Sigma = (R S)(R S)^T, f32, summed k=0,1,2."""
    c0, c1, c2 = quat_to_cols(rot)
    m0, m1, m2 = (c0 * F(scale[0])).astype(F), (c1 * F(scale[1])).astype(F), (c2 * F(scale[2])).astype(F)

    def sig(i, j):
        return F(F(F(m0[i] * m0[j]) + F(m1[i] * m1[j])) + F(m2[i] * m2[j]))
    return np.array([sig(0, 0), sig(1, 0), sig(2, 0), sig(1, 1), sig(2, 1), sig(2, 2)], dtype=F)


def cov6_f64(rot, scale):
    q = np.asarray(rot, dtype=np.float64)
    x, y, z, w = q
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    m = r @ np.diag(np.asarray(scale, dtype=np.float64))
    s = m @ m.T
    return np.array([s[0, 0], s[0, 1], s[0, 2], s[1, 1], s[1, 2], s[2, 2]])


def encode_sh(name, sh):
    if name == "single":   # gaussian_config.rs:39-41
        return sh.astype("<f4").tobytes()
    if name == "half":     # gaussian_config.rs:56-64: f16::from_f32 (RNE) x45 + one zero
        return np.concatenate([sh.astype(np.float16), np.zeros(1, np.float16)]).astype("<f2").tobytes()
    if name == "norm8":    # gaussian_config.rs:92-100: (v*127).clamp(-127,127) as i8, + 3 zeros
        v = np.clip((sh * F(127.0)).astype(F), F(-127.0), F(127.0))
        return np.concatenate([np.trunc(v).astype(np.int8), np.zeros(3, np.int8)]).tobytes()
    return b""


def encode_cov(name, rot, scale):
    if name == "rot_scale":  # gaussian_config.rs:173-175
        return np.concatenate([rot, scale]).astype("<f4").tobytes()
    c6 = cov6_f32(rot, scale)
    if name == "single":
        return c6.astype("<f4").tobytes()
    return c6.astype(np.float16).astype("<f2").tobytes()  # gaussian_config.rs:226-228


def pack(sh_name, cov_name, g):
    """src/buffer/gaussian.rs:305-339: pos, color, sh, cov3d, zero padding."""
    b = g["pos"].astype("<f4").tobytes() + g["color"].tobytes()
    b += encode_sh(sh_name, g["sh"]) + encode_cov(cov_name, g["rot"], g["scale"])
    b += b"\0" * (4 * PADDING[(sh_name, cov_name)])
    assert len(b) == 16 + SH_BYTES[sh_name] + COV_BYTES[cov_name] + 4 * PADDING[(sh_name, cov_name)]
    assert len(b) % 16 == 0
    return b


def decode_sh(name, g):
    """gaussian.wesl:29-77 expressed on the stored elements (45 floats)."""
    if name == "single":
        return g["sh"].copy()
    if name == "half":       # unpack2x16float: exact widening
        return g["sh"].astype(np.float16).astype(F)
    if name == "norm8":      # unpack4x8snorm: max(i8/127, -1)
        v = np.clip((g["sh"] * F(127.0)).astype(F), F(-127.0), F(127.0))
        i8 = np.trunc(v).astype(np.int8)
        return np.maximum(i8.astype(F) / F(127.0), F(-1.0)).astype(F)
    return np.zeros(45, dtype=F)


def decode_cov(name, g):
    """gaussian.wesl:80-149."""
    c6 = cov6_f32(g["rot"], g["scale"])
    if name == "half":
        return c6.astype(np.float16).astype(F)
    return c6


def model_matrices(pos, rot, scale, p):
    """model_transform.wesl:13-143 evaluated in float64 (expectation with tolerance)."""
    x, y, z, w = (float(v) for v in rot)
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    s = np.diag(np.asarray(scale, dtype=np.float64))
    sr = r @ s
    inv = np.diag(1.0 / np.asarray(scale, dtype=np.float64)) @ r.T
    m = np.eye(4)
    m[:3, :3] = sr
    m[:3, 3] = pos
    world = m @ np.array([p[0], p[1], p[2], 1.0])
    # column-major flattening, as the WESL constructors / the oracle's out[4*c+r]
    return m.T.reshape(-1), sr.T.reshape(-1), inv.T.reshape(-1), world


def parse_model_ply(path):
    """ply.rs:292-384 fast path: header then count x 62 little-endian f32."""
    raw = open(path, "rb").read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    count = int([l for l in raw[:end].split(b"\n") if l.startswith(b"element vertex")][0].split()[-1])
    body = np.frombuffer(raw[end:end + count * 248], dtype="<f4").reshape(count, 62)
    return end, body


def from_ply_f64(row):
    """src/gaussian.rs:70-92 (expectation; colour compared with +-1 tolerance)."""
    pos = row[0:3]
    dc = row[6:9].astype(np.float64)
    rest = row[9:54]
    opacity = float(row[54])
    scale = np.exp(row[55:58].astype(np.float64))
    rot_wxyz = row[58:62].astype(np.float64)
    q = np.array([rot_wxyz[1], rot_wxyz[2], rot_wxyz[3], rot_wxyz[0]])
    q = q / np.linalg.norm(q)
    rgb = np.clip((dc * 0.2820948 + 0.5) * 255.0, 0, 255)
    with np.errstate(over="ignore"):
        a = np.clip(255.0 / (1.0 + np.exp(-opacity)), 0, 255)
    sh = np.stack([rest[0:15], rest[15:30], rest[30:45]], axis=1).reshape(-1)
    return pos, q, np.concatenate([rgb, [a]]), sh, scale


def main():
    out = {}
    gs = [given_gaussian(s) for s in SEEDS] + [fixed_gaussian()]
    out["seeds"] = np.array(SEEDS + [-1], dtype=np.int64)  # -1 = fixed unit-test Gaussian
    out["rot"] = np.stack([g["rot"] for g in gs])
    out["pos"] = np.stack([g["pos"] for g in gs])
    out["color"] = np.stack([g["color"] for g in gs])
    out["sh"] = np.stack([g["sh"] for g in gs])
    out["scale"] = np.stack([g["scale"] for g in gs])
    for sh in SH_NAMES:
        out[f"unpack_sh_{sh}"] = np.stack([decode_sh(sh, g) for g in gs])
        for cov in COV_NAMES:
            blob = b"".join(pack(sh, cov, g) for g in gs)
            out[f"pod_{sh}_{cov}"] = np.frombuffer(blob, dtype=np.uint8).copy()
    for cov in COV_NAMES:
        out[f"unpack_cov_{cov}"] = np.stack([decode_cov(cov, g) for g in gs])
    out["cov6_f64"] = np.stack([cov6_f64(g["rot"], g["scale"]) for g in gs])
    out["unpack_color"] = np.stack([g["color"].astype(F) / F(255.0) for g in gs])

    # transform flags: gaussian_transform.rs:63-77,178-194 and gaussian_transform.wesl:14-31
    rows = []
    for mode in range(3):
        for deg in range(4):
            for no_sh0 in (0, 1):
                for std in (0.0, 1.5, 2.0, 3.0):
                    u8 = int(F(F(std) / F(3.0)) * F(255.0))
                    flags = mode | (deg << 8) | (no_sh0 << 16) | (u8 << 24)
                    rows.append((mode, deg, no_sh0, std, u8, flags, float(F(F(u8) / F(255.0)) * F(3.0))))
    out["flags_table"] = np.array(rows, dtype=np.float64)

    # model matrices: tests/shader/model_transform.rs:100-201 case + identity
    ry = np.array([0, np.sin(np.pi / 8), 0, np.cos(np.pi / 8)])
    rx = np.array([np.sin(np.pi / 12), 0, 0, np.cos(np.pi / 12)])

    def qmul(a, b):
        ax, ay, az, aw = a
        bx, by, bz, bw = b
        return np.array([aw * bx + ax * bw + ay * bz - az * by,
                         aw * by - ax * bz + ay * bw + az * bx,
                         aw * bz + ax * by - ay * bx + az * bw,
                         aw * bw - ax * bx - ay * by - az * bz])
    cases = [((5.0, 10.0, 15.0), qmul(ry, rx).astype(F), (2.0, 3.0, 4.0), (1.0, 2.0, 3.0)),
             ((0.0, 0.0, 0.0), np.array([0, 0, 0, 1], dtype=F), (1.0, 1.0, 1.0), (1.0, 2.0, 3.0))]
    mm = []
    for pos, rot, scale, p in cases:
        m, sr, inv, world = model_matrices(pos, rot, scale, p)
        mm.append(np.concatenate([pos, rot.astype(np.float64), scale, p, m, sr, inv, world]))
    out["model_cases"] = np.stack(mm)  # [pos3, rot4, scale3, p3, mat16, sr9, inv9, world4]

    # model.ply (reference example data): decoded table
    hdr, body = parse_model_ply(os.path.join(HERE, "model.ply"))
    out["ply_header_bytes"] = np.array([hdr], dtype=np.int64)
    out["ply_body"] = body.copy()
    dec = [from_ply_f64(r) for r in body]
    out["ply_pos"] = np.stack([d[0] for d in dec])
    out["ply_rot_xyzw"] = np.stack([d[1] for d in dec])
    out["ply_color_f64"] = np.stack([d[2] for d in dec])
    out["ply_sh"] = np.stack([d[3] for d in dec])
    out["ply_scale"] = np.stack([d[4] for d in dec])

    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
    # the reference's own known answers (tests/shader/gaussian.rs, seed 42) as a sanity print
    g42 = gs[SEEDS.index(42)]
    print("seed 42 color", g42["color"], "sh[0]", g42["sh"][:3], "sh[3]", g42["sh"][9:12])
    print("seed 42 cov6 f32", cov6_f32(g42["rot"], g42["scale"]))


if __name__ == "__main__":
    main()
