"""Host-side start-up helpers of round 6 (CPU): dpr_scale_amd.configure_runtime() -- the two runtime switches are explicit, an
explicit environment value wins, "too late" is reported instead of silently ignored -- and dist.probe_watchdog -- a collective start-up
probe that does not come back ends the process with a message and exit code 70 instead of hanging."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, env=None):
    e = dict(os.environ)
    for k in ("HIP_FORCE_DEV_KERNARG", "TORCH_NCCL_HIGH_PRIORITY", "DPRHOT_RUNTIME_DEFAULTS"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + textwrap.dedent(code)], capture_output=True, text=True,
                          timeout=120, env=e)


def test_importing_the_package_does_not_touch_the_environment_and_configure_runtime_does():
    r = _run("""
        import os
        import dpr_scale_amd
        assert "HIP_FORCE_DEV_KERNARG" not in os.environ and "TORCH_NCCL_HIGH_PRIORITY" not in os.environ
        out = dpr_scale_amd.configure_runtime()
        assert out == {"HIP_FORCE_DEV_KERNARG": "set", "TORCH_NCCL_HIGH_PRIORITY": "set"}, out
        assert os.environ["HIP_FORCE_DEV_KERNARG"] == "1" and os.environ["TORCH_NCCL_HIGH_PRIORITY"] == "1"
        print("ok")
    """)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-800:]


def test_an_explicit_value_wins_and_the_switch_can_be_disabled():
    r = _run("""
        import os
        import dpr_scale_amd
        out = dpr_scale_amd.configure_runtime()
        assert out["HIP_FORCE_DEV_KERNARG"] == "kept 0" and os.environ["HIP_FORCE_DEV_KERNARG"] == "0", out
        print("ok")
    """, {"HIP_FORCE_DEV_KERNARG": "0"})
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-800:]
    r = _run("""
        import os
        import dpr_scale_amd
        assert dpr_scale_amd.configure_runtime() == {} and "HIP_FORCE_DEV_KERNARG" not in os.environ
        print("ok")
    """, {"DPRHOT_RUNTIME_DEFAULTS": "0"})
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-800:]


def test_a_process_group_that_already_exists_is_reported_not_ignored():
    r = _run("""
        import os, warnings
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29781"
        dist.init_process_group("gloo", rank=0, world_size=1)
        import dpr_scale_amd
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            out = dpr_scale_amd.configure_runtime()
        assert out["TORCH_NCCL_HIGH_PRIORITY"] == "too late" and "TORCH_NCCL_HIGH_PRIORITY" not in os.environ, out
        assert any("TORCH_NCCL_HIGH_PRIORITY" in str(x.message) for x in w)
        dist.destroy_process_group()
        print("ok")
    """)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-800:]


def test_probe_watchdog_ends_a_stuck_probe_with_a_message_and_exit_code_70():
    r = _run("""
        import time
        from dpr_scale_amd import dist as D
        with D.probe_watchdog(0.5, "test probe"):
            pass                      # comes back in time: nothing happens
        with D.probe_watchdog(0, "never armed"):
            time.sleep(0.2)
        with D.probe_watchdog(0.5, "test probe"):
            time.sleep(30)            # a rank waiting in a collective its peers never issued
        print("not reached")
    """)
    assert r.returncode == 70 and "test probe did not come back" in r.stderr and "not reached" not in r.stdout, (r.returncode, r.stderr[-500:])
