"""RCCL executed for real: every RCCL call site of the N > 1 path - `init_process_group("nccl", device_id=...)` with the
bench's timeout handling, the probe all-reduce, the device-side padded all_gathers of `gather_members` / `gather_work`, `roll_out_sharded`,
`torch.cuda.nccl.version()` - runs in ONE child process as a world-size-1 group on the one GPU of the test box and must equal the no-group
path bit for bit (evaluate/pred_rollout.py:358,398-403 is what those call sites replace).  It says nothing about xGMI."""
import pytest

pytestmark = pytest.mark.gpu


def test_world_size_one_nccl_group_runs_every_collective_call_site():
    from benchlib import rccl_world1

    out = rccl_world1.run(timeout=420.0)
    assert out.get("ok"), out
    assert out["backend"] == "nccl" and out["world_size"] == 1
    assert out["rccl_version"] and out["rccl_version"][0].isdigit()
    eq = out["bit_equal_to_no_group"]
    assert set(eq) == {"members", "work", "work_chunked", "rollout", "stats_all_gather"} and all(eq.values()), eq
    print(f"\nRCCL {out['rccl_version']} at world size 1: init {out['init_seconds']} s, collectives + tiny sharded rollout {out['collectives_seconds']} s, child {out['child_seconds']} s")
