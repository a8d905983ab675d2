"""`bench.py --gpus N` as the driver issues it: the parent is a launcher that starts N ranks (one per GPU) as a child
`torch.distributed.run`, relays rank 0's JSON line as the last stdout line and fails when a rank fails -- the role Lightning's
DDPPlugin plays upstream (train.py:51-53,73-78).  No GPU here: the plan is checked dry, and the process plumbing with gloo ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e,
                          timeout=timeout)


def test_dry_run_plan_is_one_rank_per_gpu_on_loopback():
    r = _run("--gpus", "8", "--steps", "7", "--warmup", "2", "--dry-run")
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = plan["cmd"]
    assert plan["n_ranks"] == 8
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "7", "--warmup", "2"]       # the ranks get the same flags, minus the launcher's own
    assert plan["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and plan["env"]["WG_BENCH_LAUNCHED"] == "1"


def test_plan_function_strips_launcher_flags():
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args(["--gpus", "2", "--spawn", "--no-cpu"])
    cmd, env = bench.launcher_plan(args, ["--gpus", "2", "--spawn", "--no-cpu"], port=12345)
    assert "--spawn" not in cmd and cmd[-3:] == ["--gpus", "2", "--no-cpu"] and "12345" in cmd
    assert int(env["OMP_NUM_THREADS"]) >= 1


def test_launcher_starts_ranks_and_relays_rank0_line():
    r = _run("--gpus", "2", "--steps", "4", "--warmup", "1", "--selftest", "ok")
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    line = json.loads(last)                                                       # the JSON line is the LAST thing on stdout
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["value"] == 2.0 and line["steps"] == 4 and line["launched"]


def test_single_rank_through_the_launcher():
    r = _run("--gpus", "1", "--spawn", "--selftest", "ok")
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_failing_rank_fails_the_bench():
    r = _run("--gpus", "2", "--selftest", "fail")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]            # no result line from a failed run
    assert "exited with code" in r.stderr


def test_world_size_must_match_gpus():
    r = _run("--gpus", "4", "--selftest", "ok", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.parametrize("model", ["waveflow", "wsrglow"])
def test_model_flag_reaches_every_rank(model):
    """BASELINE.json configs[3] / [4] (`train.py:51-53,73-78` trains any config on N GPUs): `--model` goes through the same launcher."""
    r = _run("--model", model, "--gpus", "4", "--dry-run")
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["cmd"]
    assert cmd[cmd.index(BENCH) + 1:] == ["--model", model, "--gpus", "4"]
    r = _run("--model", model, "--gpus", "2", "--selftest", "ok")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["model"] == model and line["n_gpus"] == 2


def test_unknown_model_is_refused():
    r = _run("--model", "melglow", "--dry-run")
    assert r.returncode != 0


def test_launch_site_names_match_rocprof_names_by_prefix(tmp_path, monkeypatch):
    """roofline.traffic is looked up under the kernel that RAN (wg_timer_read_name -> launch_site_to_rocprof), by prefix against the
    names rocprofv3 prints: defaulted template arguments and the `void` / argument list are rocprofv3's, not the launch site's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.launch_site_to_rocprof("(convgemm16q_kernel<EPI_GATE_SO, 2, 2>)") == "convgemm16q_kernel<5, 2, 2"
    assert bench.launch_site_to_rocprof("convgemm16g_kernel<EPI_GATE_SO>") == "convgemm16g_kernel<5"
    assert bench.launch_site_to_rocprof("(convgemm16q_kernel<EPI_GATE, 2, 2, false, true>)") == "convgemm16q_kernel<1, 2, 2, false, true"
    assert bench.launch_site_to_rocprof("wgrad16t_kernel") == "wgrad16t_kernel"
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r09a_hbm_traffic.json").write_text(json.dumps({"kernels": {
        "void convgemm16q_kernel<5, 2, 2, false, false>(ConvGemm16sArgs)": {"hbm_bytes_per_launch": 2.0e8},
        "void convgemm16q_kernel<5, 2, 1, false, false>(ConvGemm16sArgs)": {"hbm_bytes_per_launch": 9.0e8},
        "void convgemm16q_kernel<5, 2, 2, false, true>(ConvGemm16sArgs)": {"hbm_bytes_per_launch": 7.0e8},
        "void convgemm16g_kernel<5>(ConvGemm16sArgs)": {"hbm_bytes_per_launch": 1.9e8},
        "convlayer16g_kernel": {"hbm_bytes_per_launch": 4.1e8}}}))
    (prof / "r09a_wsr_hbm_traffic.json").write_text(json.dumps({"kernels": {
        "void convgemm16q_kernel<5, 1, 1, false, false>(ConvGemm16sArgs)": {"hbm_bytes_per_launch": 3.0e8},
        "void convgemm16g_kernel<8>(ConvGemm16sArgs)": {"hbm_bytes_per_launch": 1.0e8},
        "gate_finish16g_kernel": {"hbm_bytes_per_launch": 0.5e8}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    t, src, full = bench._traffic("convgemm16q_kernel<5, 2, 2")
    assert t == 2.0e8 and src.endswith("r09a_hbm_traffic.json") and full.startswith("convgemm16q_kernel<5, 2, 2, false, false>")
    t, src, full = bench._traffic("convgemm16g_kernel<5")
    assert t == 1.9e8 and full == "convgemm16g_kernel<5>(ConvGemm16sArgs)"
    # (the site that leaves out `false, false` must not be served the `false, true` instantiation's bytes, whichever comes last)
    assert bench.full_instantiation("convgemm16q_kernel<5, 2, 2") == "convgemm16q_kernel<5, 2, 2, false, false>"
    assert bench.full_instantiation("convgemm16q_kernel<5, 2") == "convgemm16q_kernel<5, 2, 1, false, false>"
    assert bench._traffic("convgemm16q_kernel<5, 2, 2, false, true")[0] == 7.0e8
    assert bench._traffic("convlayer16g_kernel")[0] == 4.1e8                # (no template arguments: the whole name)
    assert bench._traffic("convgemm16q_kernel<1, 2, 2")[0] is None        # (a kernel no committed summary holds: nothing is cited)
    assert bench._traffic("convgemm16q_kernel<5, 1, 1", "_wsr_")[0] == 3.0e8
    # one timed class entry that covers two launches (the gate conv cut along K and the kernel that finishes it): both names, both bytes
    t, src, full = bench.site_traffic("convgemm16g_kernel<WGG_EPI_PART> + gate_finish16g_kernel", "_wsr_")
    assert t == 1.5e8 and src.endswith("r09a_wsr_hbm_traffic.json")
    assert full == "convgemm16g_kernel<8>(ConvGemm16sArgs) + gate_finish16g_kernel"
    assert bench.site_traffic("convgemm16g_kernel<WGG_EPI_PART> + gate_finish16g_kernel")[0] is None      # (the WaveGlow summary holds neither)
    assert bench.site_traffic("convlayer16g_kernel")[0] == 4.1e8
