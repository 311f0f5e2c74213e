#!/usr/bin/env python3
"""Golden vectors for the SPZ source format, from an independent vectorised numpy restatement of
src/source_format/spz.rs:436-794 and src/gaussian.rs:126-352 (it shares no code with oracle/ or the
product).  Inputs: tests/golden/model.spz (the reference's examples/model.spz data file: 9 points,
version 2, SH degree 3, 12 fractional bits) and the seeded fixture Gaussians of golden_v1.npz.
Output: tests/golden/golden_spz_v1.npz.   Run:  python tests/golden/make_golden_spz.py
f32 arithmetic throughout; exp/ln go through numpy's float32 kernels, so `scale` may sit 1 ulp from
libm — the tests compare that column with a 2-ulp budget and every byte column exactly."""
import gzip
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32
A_B = f32(0.2820948) / f32(0.15)                 # gaussian.rs:126-127
C0 = (f32(1.0) - A_B) * (f32(0.5) * f32(255.0))  # gaussian.rs:129-130
NCOEF = {0: 0, 1: 3, 2: 8, 3: 15}


def rust_round(x):
    """f32::round — half away from zero"""
    x = np.asarray(x, dtype=f32)
    return (np.sign(x) * np.floor(np.abs(x) + f32(0.5))).astype(f32)


def decode(raw):
    """decompressed payload -> dict of columns as Gaussian::from_spz produces them"""
    raw = np.frombuffer(raw, dtype=np.uint8)
    magic, version, n = np.frombuffer(raw[:12].tobytes(), dtype="<u4")
    deg, frac, flags = int(raw[12]), int(raw[13]), int(raw[14])
    assert magic == 0x5053474E and 1 <= version <= 3 and deg <= 3
    n = int(n)
    nc = NCOEF[deg]
    pb = 6 if version == 1 else 9
    rb = 4 if version >= 3 else 3
    off = 16
    cols = {}
    for name, width in (("positions", pb), ("alphas", 1), ("colors", 3), ("scales", 3), ("rotations", rb),
                        ("shs", 3 * nc)):
        cols[name] = raw[off:off + n * width].reshape(n, width)
        off += n * width
    if version == 1:
        pos = np.frombuffer(cols["positions"].tobytes(), dtype="<f2").reshape(n, 3).astype(f32)
    else:
        p = cols["positions"].reshape(n, 3, 3).astype(np.int64)
        fixed = p[..., 0] | (p[..., 1] << 8) | (p[..., 2] << 16)
        fixed = np.where(fixed & 0x800000, fixed - (1 << 24), fixed)
        pos = fixed.astype(f32) * (f32(1.0) / f32(1 << frac))
    scale = np.exp(cols["scales"].astype(f32) / f32(16.0) - f32(10.0)).astype(f32)
    if version < 3:
        xyz = cols["rotations"].astype(f32) / f32(127.5) - f32(1.0)
        l2 = (xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1]) + xyz[:, 2] * xyz[:, 2]
        w = np.sqrt(np.maximum(f32(1.0) - l2, f32(0.0)))
        rot = np.concatenate([xyz, w[:, None]], axis=1).astype(f32)
    else:
        word = np.frombuffer(cols["rotations"].tobytes(), dtype="<u4").astype(np.uint64)
        big = (word >> np.uint64(30)).astype(np.int64)
        rot = np.zeros((n, 4), dtype=f32)
        acc = np.zeros(n, dtype=f32)
        for a in range(4):   # ascending, as gaussian.rs:171-190 does
            live = big != a
            field = word & np.uint64(0x3FF)
            val = (f32(np.sqrt(0.5)) * ((field & np.uint64(0x1FF)).astype(f32) / f32(511.0))
                   * np.where(field & np.uint64(0x200), f32(-1.0), f32(1.0)).astype(f32)).astype(f32)
            rot[:, a] = np.where(live, val, f32(0.0))
            acc = np.where(live, acc + val * val, acc).astype(f32)
            word = np.where(live, word >> np.uint64(10), word)
        rot[np.arange(n), big] = np.sqrt(np.maximum(f32(1.0) - acc, f32(0.0)))
    rgb = np.clip(cols["colors"].astype(f32) * A_B + C0, f32(0.0), f32(255.0)).astype(np.uint8)
    color = np.concatenate([rgb, cols["alphas"]], axis=1)
    sh = np.zeros((n, 45), dtype=f32)
    sh[:, :3 * nc] = (cols["shs"].astype(f32) - f32(128.0)) / f32(128.0)
    return dict(version=int(version), n=n, sh_degree=deg, fractional_bits=frac, flags=flags, pos=pos, rot=rot,
                scale=scale, color=color, sh=sh)


def encode(g, version=3, sh_degree=3, fractional_bits=12, antialiased=False, sh_bits=(5, 4, 4)):
    """fixture columns -> decompressed payload bytes (Gaussian::to_spz + write_decompressed)"""
    n = len(g["pos"])
    nc = NCOEF[sh_degree]
    hdr = np.array([0x5053474E, version, n], dtype="<u4").tobytes() + bytes(
        [sh_degree, fractional_bits, 1 if antialiased else 0, 0])
    pos = g["pos"].astype(f32)
    if version == 1:
        positions = pos.astype("<f2").tobytes()
    else:
        fixed = rust_round(pos * f32(1 << fractional_bits)).astype(np.int64)
        b = np.stack([(fixed >> s) & 0xFF for s in (0, 8, 16)], axis=-1).astype(np.uint8)
        positions = b.tobytes()
    alphas = g["color"][:, 3].astype(np.uint8).tobytes()
    colors = np.clip((g["color"][:, :3].astype(f32) - C0) / A_B, f32(0), f32(255)).astype(np.uint8).tobytes()
    scales = np.clip(rust_round((np.log(g["scale"].astype(f32)).astype(f32) + f32(10.0)) * f32(16.0)), f32(0),
                     f32(255)).astype(np.uint8).tobytes()
    r = g["rot"].astype(f32)
    ln = np.sqrt(((r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1]) + r[:, 2] * r[:, 2]) + r[:, 3] * r[:, 3]).astype(f32)
    q = (r / ln[:, None]).astype(f32)
    if version >= 3:
        mag = np.abs(q)
        big = 3 - np.argmax(mag[:, ::-1], axis=1)    # last maximum, as Iterator::max_by
        flip = q[np.arange(n), big] < 0
        word = big.astype(np.uint64)
        for a in range(4):
            live = big != a
            mf = np.clip(f32(511.0) * (mag[:, a] * f32(np.sqrt(2.0))) + f32(0.5), f32(0), f32(510.0))
            field = ((((q[:, a] < 0) ^ flip).astype(np.uint64)) << np.uint64(9)) | mf.astype(np.uint64)
            word = np.where(live, (word << np.uint64(10)) | field, word)
        rotations = word.astype("<u4").tobytes()
    else:
        s = np.where(q[:, 3] < 0, f32(-1.0), f32(1.0)).astype(f32)
        rotations = np.clip(rust_round((q[:, :3] * s[:, None] + f32(1.0)) * f32(127.5)), f32(0),
                            f32(255)).astype(np.uint8).tobytes()
    shs = b""
    if nc:
        bucket = 1 << (8 - sh_bits[sh_degree - 1])
        qv = np.maximum(rust_round(g["sh"][:, :3 * nc].astype(f32) * f32(128.0) + f32(128.0)), f32(0)).astype(
            np.int64)
        if bucket < 8:
            qv = (qv + bucket // 2) // bucket * bucket
        shs = np.clip(qv, 0, 255).astype(np.uint8).tobytes()
    return hdr + positions + alphas + colors + scales + rotations + shs


ENCODE_CASES = [
    dict(version=1), dict(version=2), dict(version=3),
    dict(sh_degree=0), dict(sh_degree=1), dict(sh_degree=2),
    dict(fractional_bits=8), dict(fractional_bits=16),
    dict(sh_bits=(0, 0, 0)), dict(sh_bits=(8, 8, 8)), dict(sh_bits=(2, 4, 6)), dict(sh_bits=(4, 5, 5)),
    dict(version=2, sh_degree=1, fractional_bits=10, antialiased=True, sh_bits=(6, 6, 6)),
]


def main():
    out = {}
    with open(os.path.join(HERE, "model.spz"), "rb") as f:
        raw = gzip.decompress(f.read())
    d = decode(raw)
    out["model_header"] = np.array([d["version"], d["n"], d["sh_degree"], d["fractional_bits"], d["flags"]])
    for k in ("pos", "rot", "scale", "color", "sh"):
        out["model_" + k] = d[k]
    z = np.load(os.path.join(HERE, "golden_v1.npz"))
    g = {k: z[k] for k in ("rot", "pos", "color", "sh", "scale")}
    g["sh"] = g["sh"].reshape(len(g["pos"]), 45)
    for i, case in enumerate(ENCODE_CASES):
        payload = encode(g, **case)
        out[f"enc{i}_bytes"] = np.frombuffer(payload, dtype=np.uint8)
        dd = decode(payload)
        for k in ("pos", "rot", "scale", "color", "sh"):
            out[f"enc{i}_{k}"] = dd[k]
    out["encode_cases"] = np.array([repr(c) for c in ENCODE_CASES])
    np.savez_compressed(os.path.join(HERE, "golden_spz_v1.npz"), **out)
    print("wrote golden_spz_v1.npz:", len(out), "arrays; model.spz header", out["model_header"])


if __name__ == "__main__":
    main()
