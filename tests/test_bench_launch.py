"""bench.py as the driver types it: `python3 bench.py --gpus N` starts its own
ranks when no launcher is around it, and the LAST stdout line is one compact JSON record that fits
the driver's tail buffer (<= 4 KB) whatever the detail record holds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _run(args, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    return res


def test_self_launch_two_ranks_gloo_launch_check():
    """No GPU needed: `bench.py --gpus 2` without a launcher around it spawns torch.distributed.run as a
    child, both ranks join a gloo group, rank 0's line is relayed as the parent's last line."""
    res = _run(["--gpus", "2", "--backend", "gloo", "--launch-check"])
    assert res.returncode == 0, res.stderr[-3000:]
    last = res.stdout.strip().splitlines()[-1]
    j = json.loads(last)
    assert j["launch_check"] and j["self_launched"]
    assert j["distributed"] == {"backend": "gloo", "world_size": 2}


def test_self_launch_eight_ranks_gloo_launch_check():
    """The command the driver types on the 8-GPU node, as far as a machine without GPUs can take it:
    eight ranks come up under the self-started launcher, join one group and rank 0's line comes back."""
    res = _run(["--gpus", "8", "--backend", "gloo", "--launch-check"])
    assert res.returncode == 0, res.stderr[-3000:]
    j = json.loads(res.stdout.strip().splitlines()[-1])
    assert j["launch_check"] and j["self_launched"]
    assert j["distributed"] == {"backend": "gloo", "world_size": 8}


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], cwd=ROOT,
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 2 and "WORLD_SIZE=3" in res.stderr


def _fake_detail(world=8):
    """A detail record at least as large as a real one (BENCH_r03's line was 20 KB)."""
    stages = {k: 0.123456789 for k in ("repack", "preprocess", "scan", "depth_sort", "expand", "tile_sort", "ranges",
                                        "blend", "frame")}
    models = {k: dict(bound="hbm", model_bytes=123456789012, model="x" * 400, ms=0.1234567, achieved_gbs=1234.5678,
                      frac=0.4567891) for k in ("preprocess", "depth_sort", "expand", "tile_sort", "ranges")}
    fb = dict(survey_model_bytes=1, survey_model="y" * 300, pmc_frame_frac=0.4912345)

    def summary():
        return dict(workload="w" * 80, value=12345.678901, unit="Msplats/s", ms_per_step=1.23456789, frame_ms=None,
                    visible=7091153, pairs=25495370, launches_per_frame=20, stages_ms=dict(stages),
                    stage_models=dict(models), frame_bytes=dict(fb), preprocess_read_frac=0.7234567,
                    per_rank_ms=[1.23456789] * world)

    roof = dict(bound="hbm", kernel="k_preprocess_banded<ShSingle,RotScale,pipelined,nt>", workload="w" * 80,
                achieved=5811.123456, peak=8000.0, unit="GB/s", frac=0.72637123, algorithmic_bytes_per_launch=2240000000,
                avg_launch_ms=0.3854812, visible=7091153, gaussians=10_000_000, traffic=2010400000.0,
                fetched_over_required=0.7542435, physical_traffic_frac=0.6519123, frac_nocull=0.5684123)
    return dict(
        metric="Msplats/s @1080p (Gaussians per second through proj+sort+blend)", value=2814.731234, unit="Msplats/s",
        n_gpus=world, steps=20, warmup=5, ms_per_step=0.35527412, higher_is_better=True, scaling="strong",
        vs_baseline=None, dtype="f32", data="synthetic",
        config=dict(workload="1M synthetic Gaussians, SH degree 0 (ShNone/RotScale 48 B), 1920x1080", gaussians=1000000,
                    visible=708615, pairs=2550276, sort_passes=5, launches_per_frame=19,
                    parallelism="tile-row bands x8 + one RCCL all-gather (tile rows re-cut to equal pairs (one calibration frame))",
                    frames_in_flight=2,
                    image_checksum=5161086.015),
        frame_ms=dict(median=0.34256123, samples=100), stages_ms=dict(stages), stage_models=dict(models), frame_bytes=dict(fb),
        in_flight_run=dict(ms_per_step=0.2749212, frames_in_flight=2, images_bit_identical=True),
        single_stream=dict(ms_per_step=0.3276212, value=3052.123456, steady_state=dict(ms_per_step=0.3031212, value=3299.1, frames=200)),
        blend=dict(note="z" * 500),
        hip_runtime=dict(source="already-mapped", compiled_version=70226015, runtime_version=70051831),
        distributed=dict(backend="nccl", world_size=world, per_rank_ms=[0.123456789] * world, per_rank_ms_min=0.1,
                         per_rank_ms_max=0.2, render_ms_per_rank=[0.123456789] * world, gather_ms_per_rank=[0.0123456789] * world,
                         bands=[[i * 8, i * 8 + 8] for i in range(world)], band_plan="tile rows re-cut to equal pairs"),
        roofline=roof, roofline_workload=summary(),
        roofline_nocull=dict(frac=0.5684, avg_launch_ms=0.49261, frame=summary()),
        workloads={"10m-4k": summary(), "50m": summary()},
        cpu_baseline=dict(value=1.2345678, unit="Msplats/s", cores=16, kind="port", ms_per_frame=812.345678,
                          hardware_threads=256, single_thread=dict(value=0.1189), per_stage_best=dict(value=1.3),
                          sample="s" * 400))


@pytest.mark.parametrize("world", [1, 8])
def test_compact_line_fits_the_drivers_tail(world):
    import bench
    line = _fake_detail(world)
    if world == 1:
        line.pop("distributed")
    assert len(json.dumps(line)) > 8000            # the detail would not fit
    c = bench.compact_line(line, "gpurun_out/bench_detail.json")
    s = json.dumps(c)
    assert len(s) <= 4096, len(s)
    # the contract's keys and the two objects the judge reads survive the budget
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_nocull", "kernel"):
        assert k in c["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    assert c["config"]["workload"].startswith("1M synthetic") and c["config"]["frames_in_flight"] == 2
    assert c["single_stream"]["ms_per_step"] > c["ms_per_step"] * 0     # the one-stream figure travels with the headline
    if world > 1:
        assert c["distributed"]["world_size"] == world and c["distributed"]["backend"] == "nccl"


def test_ranges_model_follows_the_kernel_that_ran(monkeypatch):
    """the search kernel was charged for reading all D keys (frac 1.64)."""
    import bench
    wl = bench.WORKLOADS["50m"]
    st = dict(preprocess=1.3, depth_sort=0.5, expand=0.285, tile_sort=1.133, ranges=0.0194, blend=0.155)
    res = dict(pairs=127_747_685, visible=35_000_000, stages_ms=st, sort_passes=5, pair_capacity=160_000_000)
    monkeypatch.setenv("GS3D_RANGES_IN_BLEND", "1")
    assert bench.stage_models(wl, res)["ranges"]["bound"] == "fused"       # found inside the blend: no kernel to price
    assert bench.stage_models(bench.WORKLOADS["10m-4k"], res)["ranges"]["bound"] == "fused"
    monkeypatch.delenv("GS3D_RANGES_IN_BLEND")
    assert bench.stage_models(bench.WORKLOADS["10m-4k"], res)["ranges"]["bound"] == "fused"   # the default above 16384 tiles
    m = bench.stage_models(wl, res)["ranges"]
    assert m["bound"] == "latency" and m["frac"] < 1.0 and m["model_bytes"] < 127_747_685 * 2
    res["pair_capacity"] = 1 << 20                   # below the switch: the scan kernel ran
    m = bench.stage_models(wl, res)["ranges"]
    assert m["bound"] == "hbm" and m["model_bytes"] == 127_747_685 * 2 + 8160 * 8


@pytest.mark.gpu
def test_bench_two_ranks_gloo_rehearsal_self_launched(tmp_path):
    """The whole N = 2 path of bench.py on ONE GPU: typed without a launcher, gloo instead of RCCL (one
    GPU cannot host an RCCL world of two), both ranks on device 0 — band plan, rebalancing, FramePipeline,
    the gathers, the compact line with its `distributed` object."""
    detail = tmp_path / "detail.json"
    res = _run(["--gpus", "2", "--backend", "gloo", "--force-device", "0", "--workload", "100k", "--steps", "3",
                "--warmup", "1", "--no-roofline", "--no-cpu-baseline", "--frame-samples", "4", "--timing-steps", "2",
                "--detail-out", str(detail)])
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
    last = res.stdout.strip().splitlines()[-1]
    assert len(last) <= 4096
    j = json.loads(last)
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["steps"] == 3 and j["warmup"] == 1
    assert j["config"]["frames_in_flight"] == 2 and "2 frames in flight per rank" in j["config"]["parallelism"]
    d = j["distributed"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and len(d["per_rank_ms"]) == 2 and len(d["bands"]) == 2
    full = json.loads(detail.read_text())
    assert full["distributed"]["world_size"] == 2 and "stage_models" in full


@pytest.mark.gpu
def test_bench_headline_runs_with_frames_in_flight(tmp_path):
    """the default: three renderers on three priority streams take the headline's frames in turn; the images must be
    bit-identical to the single-stream frame (bench.py raises otherwise) and the one-stream figure is reported too"""
    detail = tmp_path / "detail.json"
    res = _run(["--workload", "100k", "--steps", "6", "--warmup", "2", "--no-roofline", "--no-cpu-baseline",
                "--frame-samples", "4", "--timing-steps", "2", "--no-steady", "--detail-out", str(detail)])
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
    j = json.loads(res.stdout.strip().splitlines()[-1])
    assert j["config"]["frames_in_flight"] == 3 and j["single_stream"]["ms_per_step"] > 0 and j["steps"] == 6
    full = json.loads(detail.read_text())
    fl = full["in_flight_run"]
    assert fl["images_bit_identical"] and len(set(fl["stream_priorities"])) == 3
    assert abs(full["ms_per_step"] - fl["ms_per_step"]) < 1e-12


@pytest.mark.gpu
def test_bench_single_gpu_compact_line(tmp_path):
    detail = tmp_path / "detail.json"
    res = _run(["--workload", "100k", "--steps", "3", "--warmup", "1", "--no-roofline", "--no-cpu-baseline",
                "--frame-samples", "4", "--timing-steps", "2", "--no-in-flight", "--detail-out", str(detail)])
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
    lines = res.stdout.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) <= 4096       # ONE stdout line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["hip"]["compiled"] and j["hip"]["runtime"]
    assert j["config"]["frames_in_flight"] == 1 and "single_stream" not in j        # --no-in-flight: one stream
    assert json.loads(detail.read_text())["hip_runtime"]["runtime_version"] == j["hip"]["runtime"]
