"""CPU: `python bench.py --gpus 2` starts two ranks itself (torch.distributed.run, 127.0.0.1 rendezvous) and rank 0
prints ONE JSON line with n_gpus == 2 — the launcher / barrier / max-over-ranks path of the driver contract, rehearsed
with gloo and no kernels (`--rehearse`)."""
import json
import os
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, env=env,
                       timeout=300)
    return r


def test_gpus_2_starts_two_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 64
    assert out["steps"] == 3 and out["scaling"] == "weak" and out["rehearsal"] is True
    assert out["ms_per_step"] >= 2.0          # max over ranks: rank 1 sleeps 2 ms per step, rank 0 only 1 ms


def test_gpus_flag_must_match_the_launcher():
    r = _run(["--gpus", "2", "--rehearse"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout)


def test_pmc_profile_is_selected_by_workload_and_kernel_sources(monkeypatch):
    """roofline.traffic comes from a committed counter profile of the SAME workload taken on the SAME kernel sources
    (tools/pmc_traffic.sh records bench.csrc_digest()); anything else is reported as stale / none, never as current."""
    import glob
    sys.path.insert(0, str(ROOT))
    import bench
    import argparse
    a = argparse.Namespace(batch=32, size=256, arch="DU", act="relu", norm="bn", filters=[64, 1024], optimizer="adam",
                           precision="fp32")
    monkeypatch.setattr(bench, "csrc_digest", lambda: "not-a-tree")
    got = bench.pmc_traffic("wgrad_halo9_kernel<3, 1>", a)
    assert got["traffic"] is None and got["traffic_source"] == "stale" and got["traffic_stale_profile"].startswith("profiles/")
    # a profile that records its source digest is used when (and only when) the digest matches
    for f in sorted(glob.glob(str(ROOT / "profiles" / "*_hbm_traffic_pmc.json")), reverse=True):
        tab = json.load(open(f))
        cfg = dict(tab.get("__config__") or {})
        sha = cfg.pop("csrc_sha", None)
        if sha is None:
            continue
        a2 = argparse.Namespace(**{**vars(a), **{k: cfg[k] for k in ("batch", "size", "precision") if k in cfg}})
        if bench.workload_key(a2) != cfg:
            continue
        kernel = next(k for k in tab if k != "__config__")
        monkeypatch.setattr(bench, "csrc_digest", lambda sha=sha: sha)
        got = bench.pmc_traffic(kernel, a2)
        assert got["traffic"] == int(tab[kernel]["hbm_bytes_per_launch"]) and got["traffic_source"].startswith("profiles/")
        break
    a.size = 512                               # no profile of that workload: say so instead of borrowing another one's
    assert bench.pmc_traffic("igemm_halo_bf16w4_kernel<0, true>", a) == {"traffic": None, "traffic_source": "none"}
