"""The figures bench.py takes from FILES (profiles/r06_pmc_*.json, r06_tail_roofline.json) belong to the sources in the tree: every such file carries the
hash of deep-prove_amd/csrc/ it was collected on (tools/srchash.py) and bench.py refuses one from other sources. Checked here without a GPU: the hash
function of bench.py and of the tools agree, the committed files carry the current hash (skipped, not failed, when the sources have moved on: the bench
line then reports `traffic: null` and no `tail_roofline` until tools/r06/final.sh has run again), and tools/tail_roofline.py reproduces the committed
object from the committed member-timing summaries."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("r06_pmc_dense4m_proofs.json", "r06_pmc_sumcheck24.json", "r06_pmc_sq_bench704.json", "r06_tail_roofline.json")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(b)
    finally:
        sys.argv = argv
    return b


def _tool_hash():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from srchash import source_sha16
    finally:
        sys.path.pop(0)
    return source_sha16()


def test_bench_and_tools_hash_the_sources_alike():
    assert _bench()._source_sha16() == _tool_hash()


def test_committed_counter_files_belong_to_the_sources_in_the_tree():
    b = _bench()
    cur = b._source_sha16()
    docs = {f: json.load(open(os.path.join(ROOT, "profiles", f))) for f in FILES}
    hashes = {f: d.get("source_sha16") for f, d in docs.items()}
    assert len(set(hashes.values())) == 1, f"the counter files come from different sources: {hashes}"
    if set(hashes.values()) != {cur}:
        pytest.skip(f"deep-prove_amd/csrc/ is at {cur}, the counter files at {set(hashes.values())}: bench.py will withhold them until tools/r06/final.sh has run again")
    assert all(b._same_sources(d) for d in docs.values())
    t = b.tail_roofline()
    assert t and 0.2 < t["frac_member"] < 1.0 and 0.1 < t["frac_merged_launch"] <= t["frac_member"] and t["source_sha16"] == cur


def test_tail_roofline_tool_reproduces_the_committed_object(tmp_path):
    out = tmp_path / "tail.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tail_roofline.py"), os.path.join(ROOT, "profiles", "r06_wgtimes_704_summary.txt"),
                        os.path.join(ROOT, "profiles", "r06_wgphases_704_summary.txt"), str(out)], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    got, want = json.load(open(out)), json.load(open(os.path.join(ROOT, "profiles", "r06_tail_roofline.json")))
    for k in ("permutations_per_member", "floor_us_per_member", "median_member_us", "merged_launch_us", "merged_launches", "members_per_launch", "start_skew_us",
              "slowest_minus_fastest_member_us", "frac_member", "frac_merged_launch"):
        assert got[k] == want[k], (k, got[k], want[k])
    assert abs(want["floor_us_per_member"] - want["permutations_per_member"] * want["permutation_us_in_a_loop"]) < 1.0
