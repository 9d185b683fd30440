#!/usr/bin/env python
"""bench.py — proofs/s of the MI355X-native prover on BASELINE.json's metric configs.

Headline (`value`): Dense-4M (configs[1]). A step = one batch of WAVES_PER_STEP x `concurrency` complete proofs
(zkml::Prover::prove: witness commitments, layer sumchecks, logup-GKR, table proofs, Basefold batch opening) of distinct
synthetic inputs, `concurrency` of them in flight on one GPU at any time; model weights and their commitments are resident in
HBM before the timed region (Context::generate is setup, exactly as in the reference harness zkml/src/bin/bench.rs:390-408).
The LAST timed step carries the golden input of tests/golden/*_proof.json at a non-zero index and the sha256 of that proof's
canonical stream must equal the oracle's (`golden_sha256_ok`): what is timed is bit-identical to the reference's algorithm,
not merely accepted by the verifier. The same JSON line carries CNN-264k
(configs[2]) measured the same way, the standalone 2^24 sumcheck (configs[4] on one GPU) with its HBM roofline, and the CPU
baseline (the oracle, i.e. a single-threaded port of the reference CPU path, on a bounded sample).
`config.gpu_clocks_timed_region` = shader clock / package power / busy percentage of THIS process's GPU (amdgpu hwmon, found by PCI address) sampled at 10 Hz while the
timed steps run; `roofline.peak_sustained` = the compress probe held for 1.5 s with the same sampler on.
Multi-GPU (launched by torch.distributed.run): independent proofs shard across ranks with no data-path collective
("replicas") -> weak scaling; only the timing uses a collective (MAX over ranks).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# proofs per step = WAVES_PER_STEP x proofs in flight: every call of dp_model_prove_batch starts with all cohorts in phase and ends
# with a drain (~0.2 s together, profiles/r02_waves.jsonl: 1 wave 327, 2 waves 352, 4 waves 410, 8 waves 431 proofs/s), a service
# feeds its prover continuously
BATCHES_PER_STEP = 6  # (6 waves of 448 until round 5: 2 688 proofs per step; 6 of 704 since round 6: 4 224 — 4 waves measured 1 020 proofs/s where 12-wave batches give 1 108, tools/r06/call24.sh, call25.sh)
# proofs in flight per GPU, in lock-step cohorts (csrc/hip_dev.hip, struct Cohort). The library cuts the number to what fits in the
# free HBM (worker arenas are sized from the footprint of the model's first proof: 448 MB for Dense-4M since the batch-opening
# sumcheck keeps its eq tables factored — 576 MB and at most 423 in flight before). 448 against 256 in flight, same build and box,
# alternating: 466 / 511 against 452 / 448 proofs/s in the bench's own steps (profiles/r03_graph1_bench_inflight.txt), 480 against 453
# on average in single batches (profiles/r03_cohort_inflight_ab.txt).
# Round 6: worker arenas of 336 MB (1.125 x the footprint + 16 MB) let ~700 proofs fit, and the rate still grows with the number in flight once the streaming kernels are
# capped and prioritised (tools/r06/call22.sh - call24.sh: 1 047-1 066 proofs/s at 448, 1 100-1 116 at 660, 1 108 at 704 in 12-wave batches): 704 = 22 cohorts of 32.
DEFAULT_IN_FLIGHT = 704
GOLDEN = {"dense_4m": "dense4m_proof.json", "cnn_264k": "cnn264k_proof.json"}
MIN_HOST_THREADS_PER_RANK = 3  # proving threads per rank below which a multi-GPU run is flagged host-bound (each rank also wants 2 CPUs for the HIP runtime)
GOLDEN_SLOT = 7  # index inside the last timed step at which the golden input is proved
SHARDED_WATCHDOG_S = float(os.environ.get("DP_BENCH_SHARDED_WATCHDOG_S", "240"))
PUBLISHED = {"dense_4m": 1000.0 / 2335.0, "cnn_264k": 1000.0 / 1242.0}  # reference README.md:17-18 (hardware unstated)
WORKLOADS = {
    "dense_4m": "Dense-4M MLP (mlp.py --num-dense 5 --layer-width 1024: 4->1024->1024x4->3, Dense+Requant+ReLU blocks, 4.21M params), 1 input per proof",
    "cnn_264k": "CNN-264k on CIFAR-10 shapes (cifar-cnn.py --num-params 264000: conv 3->12 5x5, pool, conv 12->33 5x5, pool, fc 825->247->173->10, Requant after every conv/fc), 1 input per proof",
    "mlp_w256": "MLP 3x256 (smoke)",
}


def shard(total, world, rank):
    """indices of the proofs rank `rank` owns out of `total` (round robin)"""
    return list(range(rank, total, world))


class ClockSampler:
    """Shader clock, package power and busy percentage of the GPU sampled WHILE the timed steps run (a thread reading the amdgpu hwmon / sysfs files every
    100 ms; `rocm-smi --json` once a second when sysfs is not visible): a 1.4 kW part running 64-bit integer multiply-adds on every SIMD may not hold its
    2.4 GHz boost clock, and every "fraction of the VALU issue slots" quoted from a counter pass is only as good as the clock it assumes. Never fails the
    bench: whatever cannot be read is missing from the summary."""

    def __init__(self, device_index=0, period_s=0.1, sysfs_device=None):
        """sysfs_device: the device directory to read instead of looking this process's GPU up (tests point it at a fake tree)"""
        import glob
        self.period = period_s
        self.rows = []
        self.files = {}
        self.smi = None
        self.pci = None
        self._stop = None
        self._thread = None
        # the sysfs node of THIS process's HIP device, by PCI address (a box shows every GPU of the node in /sys, the container only owns some of them)
        cards = [sysfs_device] if sysfs_device else []
        try:
            if cards:
                raise LookupError("given")
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            if os.path.isdir(f"/sys/bus/pci/devices/{bdf}"):
                cards = [f"/sys/bus/pci/devices/{bdf}"]
                device_index = 0
                self.pci = bdf
        except Exception:  # noqa: BLE001
            pass
        if not cards and os.path.isdir("/sys/class/drm"):
            for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
                try:
                    if open(os.path.join(dev, "vendor")).read().strip() == "0x1002":
                        cards.append(dev)
                except OSError:
                    pass
            if len(cards) > 1:
                cards = []  # several GPUs in /sys and no PCI address to tell which one is ours: rather no reading than another GPU's
        if cards:
            dev = cards[0]
            cand = {"sclk_hz": ["hwmon/hwmon*/freq1_input"], "power_uw": ["hwmon/hwmon*/power1_average", "hwmon/hwmon*/power1_input"],
                    "busy_pct": ["gpu_busy_percent"], "temp_mc": ["hwmon/hwmon*/temp2_input", "hwmon/hwmon*/temp1_input"]}
            for key, pats in cand.items():
                for pat in pats:
                    hits = sorted(glob.glob(os.path.join(dev, pat)))
                    if hits:
                        try:
                            float(open(hits[0]).read().strip())
                            self.files[key] = hits[0]
                            break
                        except (OSError, ValueError):
                            pass
        if "sclk_hz" not in self.files:
            for exe in ("/opt/rocm/bin/rocm-smi", "rocm-smi"):
                if os.path.exists(exe) or exe == "rocm-smi":
                    self.smi = [exe, "-d", str(device_index), "--showclocks", "--showpower", "--showuse", "--json"]
                    break

    def _read_smi(self):
        import subprocess
        try:
            out = subprocess.run(self.smi, capture_output=True, text=True, timeout=10).stdout
            doc = json.loads(out[out.index("{"):])
            card = next(iter(doc.values()))
            row = {}
            for k, v in card.items():
                kl = k.lower()
                if "sclk" in kl and "mhz" in str(v).lower():
                    row["sclk_hz"] = 1e6 * float(str(v).lower().replace("mhz", "").strip("() "))
                elif "power" in kl and "(w)" in kl:
                    try:
                        row["power_uw"] = 1e6 * float(v)
                    except ValueError:
                        pass
                elif kl.startswith("gpu use"):
                    try:
                        row["busy_pct"] = float(v)
                    except ValueError:
                        pass
            return row
        except Exception:  # noqa: BLE001
            return {}

    def _loop(self):
        while not self._stop.is_set():
            row = {}
            if self.files:
                for key, path in self.files.items():
                    try:
                        row[key] = float(open(path).read().strip())
                    except (OSError, ValueError):
                        pass
            elif self.smi:
                row = self._read_smi()
            if row:
                self.rows.append(row)
            self._stop.wait(self.period if self.files else 2.0)

    def start(self):
        import threading
        if not self.files and not self.smi:
            return self
        self.rows = []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=15)
            self._thread = None
        return self.summary()

    def summary(self):
        if not self.rows:
            return None

        def col(key, scale):
            v = sorted(r[key] * scale for r in self.rows if key in r)
            return None if not v else {"mean": round(sum(v) / len(v), 1), "min": round(v[0], 1), "median": round(v[len(v) // 2], 1), "max": round(v[-1], 1)}

        return {"samples": len(self.rows), "source": f"amdgpu hwmon (sysfs{', PCI ' + self.pci if self.pci else ''}), 10 Hz" if self.files else "rocm-smi --json, 0.5 Hz", "sclk_mhz": col("sclk_hz", 1e-6),
                "power_w": col("power_uw", 1e-6), "busy_pct": col("busy_pct", 1.0), "temp_c": col("temp_mc", 1e-3)}


def timed_region(prove_batch, my_inputs, conc, steps, warmup, dist=None, device_sync=None, reduce_device="cpu"):
    """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + device sync on both sides; returns
    (MAX over ranks of the elapsed seconds, result of the last step). A step = one batch of `conc` proofs (`conc` here is
    the batch size of a step; how many of them are in flight at once is the prover's business)."""
    import torch

    def barrier():
        if dist is not None:
            dist.barrier()
        if device_sync is not None:
            device_sync()

    for i in range(warmup):
        prove_batch(my_inputs[i * conc:(i + 1) * conc])
    barrier()
    sampler = ClockSampler(int(os.environ.get("LOCAL_RANK", "0"))).start()
    t0 = time.perf_counter()
    last = None
    marks = [t0]
    for i in range(steps):
        lo = (warmup + i) * conc
        last = None  # a step's proofs are 6 MB each: release them before the next step allocates its own
        last = prove_batch(my_inputs[lo:lo + conc])
        marks.append(time.perf_counter())  # (diagnostics only: the measurement is the bracket around all K steps)
    barrier()
    elapsed = time.perf_counter() - t0
    timed_region.clocks = sampler.stop()
    timed_region.step_ms = [round(1000 * (b - a), 1) for a, b in zip(marks, marks[1:])]
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    return elapsed, last


def guarded(fn, seconds, on_timeout):
    """fn() under a watchdog: its result, {"error": ...} if it raises, and `on_timeout()` (from a timer thread) if it has not
    returned after `seconds` — a collective that never returns cannot be cancelled, only left behind"""
    import threading
    dog = threading.Timer(seconds, on_timeout)
    dog.daemon = True
    dog.start()
    try:
        return fn()
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        dog.cancel()


def make_model(dpa, workload):
    return {"dense_4m": dpa.models.dense_4m, "cnn_264k": dpa.models.cnn_264k, "mlp_w256": lambda: dpa.models.mlp(3, 256, config=5)}[workload]()


def strong_share(batch, world, rank):
    """BASELINE config 4 ("Dense 4M, batch of 64 independent proofs sharded 8 x MI355X"): the proofs of ONE fixed batch that rank
    `rank` proves — a contiguous-by-stride partition, sizes differ by at most one"""
    return len(shard(batch, world, rank))


def transformer_layer_section(dpa, dev, conc=320):  # (round 3: 64 in flight 87 proofs/s, 192: 142; round 4: 192: 163, 320: 186 — profiles/r04_transformer_layer_sweeps.txt)
    """SURVEY §8 f4 measured: one whole pre-LN transformer layer as one graph of 19 nodes (models.transformer_layer: LayerNorm, QKV, the Mha node of
    transformer/mha.rs, projection, residual; LayerNorm, Linear, ReLU, Linear, residual) AT THE SIZE THE GOLDEN PINS: case 14 of
    tests/golden/graph_models.json = the oracle's sha256 at 64 tokens x 256 features, 4 heads of 64, ffn 1024, config 66. The golden input sits in a slot
    of the LAST timed batch: its throughput-mode proof must have the oracle's sha256, and EVERY proof of that batch must verify (as measure_workload does
    for Dense-4M / CNN-264k)."""
    import hashlib
    import numpy as np
    with open(os.path.join(ROOT, "tests", "golden", "graph_models.json")) as f:
        c = json.load(f)[14]
    assert c["model"] == "transformer_layer" and c["args"]["seq"] == 64 and c["args"]["emb"] == 256
    a = c["args"]
    g = dpa.models.transformer_layer(**a)
    ctx = dpa.Context.generate(dev, g.blob())
    pr = dpa.Prover(ctx)
    x = g.input()
    assert hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest() == c["input_sha256"]
    proof, out = pr.prove(x)
    latency_golden_ok = bool(proof.size == c["proof_words"] and hashlib.sha256(proof.tobytes()).hexdigest() == c["proof_sha256"])
    assert latency_golden_ok, "transformer layer (64 x 256): the latency-mode proof of the golden input differs from the oracle's proof stream"
    dpa.verify(ctx.verifier_blob(), proof, x, out)
    lat = []
    for _ in range(3):
        t0 = time.perf_counter(); pr.prove(x); lat.append(1000 * (time.perf_counter() - t0))
    nb = 3  # batches of `conc` proofs in the timed call; the golden input rides in the last one
    xs = np.stack([g.input(100 + i) for i in range(nb * conc)])
    gslot = (nb - 1) * conc + min(GOLDEN_SLOT, conc - 1)
    xs[gslot] = x
    pr.prove_batch(xs[:conc], conc)
    sampler = ClockSampler(int(os.environ.get("LOCAL_RANK", "0"))).start()
    t0 = time.perf_counter(); proofs, outs, _ = pr.prove_batch(xs, conc); dt = time.perf_counter() - t0
    clocks = sampler.stop()
    gp = proofs[gslot]
    golden_ok = bool(gp.size == c["proof_words"] and hashlib.sha256(gp.tobytes()).hexdigest() == c["proof_sha256"]
                     and hashlib.sha256(np.ascontiguousarray(outs[gslot]).tobytes()).hexdigest() == c["output_sha256"])
    assert golden_ok, "transformer layer (64 x 256): the throughput-mode proof of the golden input differs from the oracle's proof stream"
    lo = (nb - 1) * conc
    v, vms = dpa.verify_batch(ctx.verifier_blob(), proofs[lo:], xs[lo:], outs[lo:], dev=dev)
    assert not v.any(), f"transformer layer: {int((v != 0).sum())} of {conc} proofs of the last batch were rejected"
    r = {"value": round(len(xs) / dt, 2), "unit": "proofs/s", "workload": f"one pre-LN transformer layer, {len(g.nodes)} nodes, seq {a['seq']} x emb {a['emb']}, {a['heads']} heads of {a['head_dim']}, ffn {a['ffn']} (Mha as one node; ReLU for GELU) = golden case 14",
         "proofs": len(xs), "proofs_in_flight": int(pr.in_flight()), "single_proof_latency_ms": round(sorted(lat)[1], 2), "proof_words": int(proof.size),
         "golden_sha256_ok": golden_ok, "golden_sha256_ok_latency_mode": latency_golden_ok, "verified": int(conc), "rejected": int(v.sum()), "verify_batch_ms_per_proof": round(vms / conc, 3), "gpu_clocks_timed_region": clocks}
    ctx.free()
    return r


def measure_workload(dpa, dev, workload, conc, steps, warmup, world, rank, dist, torch, profile=False, strong_batch=0):
    """setup + latency of one proof + the timed throughput region + verification of the last batch (+ the per-kernel HIP
    event profile of one more proof); the model context and its workers are released before returning"""
    import numpy as np
    mb = make_model(dpa, workload)
    t0 = time.time()
    ctx = dpa.Context.generate(dev, mb.blob())  # setup: weight commitments (not part of proving time)
    setup_s = time.time() - t0
    prover = dpa.Prover(ctx)
    vblob = ctx.verifier_blob()
    if strong_batch:  # strong scaling: a step = this rank's share of ONE fixed batch, all of it in flight at once
        batch = strong_share(strong_batch, world, rank)
        conc = max(1, min(conc, batch))
        per_rank = (steps + warmup) * batch
        my_inputs = np.stack([mb.input(1000 + (s * strong_batch) + j) for s in range(steps + warmup) for j in shard(strong_batch, world, rank)]) if batch else np.zeros((0, mb.input(0).size), dtype=np.int64)
    else:
        batch = BATCHES_PER_STEP * conc  # proofs per step: several waves of `conc` in flight, so the ramp and drain of a step weigh less
        per_rank = (steps + warmup) * batch
        my_inputs = np.stack([mb.input(1000 + i) for i in shard(world * per_rank, world, rank)])
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", GOLDEN[workload]))) if workload in GOLDEN else None
    lo_last = (warmup + steps - 1) * batch
    gslot = min(GOLDEN_SLOT, batch - 1)
    if gold is not None and batch:
        my_inputs[lo_last + gslot] = mb.input(gold["input_index"])
    t0 = time.perf_counter()
    prover.prove(my_inputs[0])
    first_ms = 1000 * (time.perf_counter() - t0)
    lat = []
    for _ in range(3):  # the single-proof figure is the median of three (one sample caught a 40 ms host hiccup in round 2)
        t0 = time.perf_counter()
        prover.prove(my_inputs[0])
        lat.append(1000 * (time.perf_counter() - t0))
    latency_ms = sorted(lat)[1]
    cuda = torch.cuda.is_available() and (dist is None or dist.get_backend() == "nccl")
    elapsed, last = timed_region(lambda xs: prover.prove_batch(xs, conc), my_inputs, batch, steps, warmup, dist,
                                 torch.cuda.synchronize if cuda else None, "cuda" if cuda else "cpu")
    step_ms = list(getattr(timed_region, "step_ms", []))
    clocks = getattr(timed_region, "clocks", None)
    # EVERY proof of the last step must verify — an invalid proof voids the measurement. One proof through the host-only
    # verifier (dp_verify, the latency figure: protocol checks on one thread, its Merkle paths on up to 8, DP_VERIFY_THREADS), then the whole step through dp_verify_batch: protocol checks on the host
    # threads, the Merkle paths of each proof (125 000 compress() for Dense-4M) authenticated on the GPU in one launch.
    lo = (warmup + steps - 1) * batch
    checked, one, vb_ms = 0, 0.0, 0.0
    if batch:
        t0 = time.perf_counter()
        dpa.verify(vblob, last[0][0], my_inputs[lo], last[1][0])
        one = max(time.perf_counter() - t0, 1e-4)
        verdicts, vb_ms = dpa.verify_batch(vblob, last[0], my_inputs[lo:lo + batch], last[1], dev=dev)
        assert not verdicts.any(), f"{workload}: {int((verdicts != 0).sum())} of {batch} proofs of the last step were rejected"
        checked = batch
    golden_ok = None
    if gold is not None and batch:  # bit identity of what was timed: the proof of the golden input out of the last timed step (throughput mode)
        import hashlib
        g = last[0][gslot]
        golden_ok = bool(g.size == gold["proof_words"] and hashlib.sha256(g.tobytes()).hexdigest() == gold["sha256"] and [int(v) for v in last[1][gslot]] == gold["output"])
        assert golden_ok, f"{workload}: the throughput-mode proof of the golden input differs from the oracle's proof stream"
    in_flight = prover.in_flight()
    rep = kernel_profile(dev, prover, my_inputs[0]) if profile else None
    ctx.free()  # releases the workers' arenas too: the next workload sizes its own against the free HBM
    return dict(mb=mb, elapsed=elapsed, latency_ms=latency_ms, latency_samples_ms=[round(v, 2) for v in lat], first_ms=first_ms, setup_s=setup_s, proof_words=int(last[0][0].size),
                verified=checked, step_ms=step_ms, verify_ms=round(1000 * one, 2), verify_batch_ms_per_proof=round(vb_ms / max(1, checked), 3), in_flight=in_flight, kernel_report=rep, golden_ok=golden_ok, clocks=clocks)


def cnn_steps(steps):
    """timed steps of the CNN-264k section of a Dense-4M run (the headline keeps the K the caller asked for)"""
    return max(1, min(steps - 1, 5))


def kernel_profile(dev, prover, x):
    """HIP-event timing of every kernel of ONE proof on the launch stream (untimed extra proof)"""
    dev.profile(True)
    prover.prove(x)
    rep = dev.profile_report()
    dev.profile(False)
    rep.sort(key=lambda r: -r["total_ms"])
    return rep


def _source_sha16():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from srchash import source_sha16
    return source_sha16()


def _same_sources(doc):
    """a committed counter pass / diagnostic timing describes THIS build only if it was collected on the same deep-prove_amd/csrc (tools/srchash.py); files from
    before round 5 carry no hash and are never quoted"""
    return bool(doc.get("source_sha16")) and doc["source_sha16"] == _source_sha16()


def valu_accounting(workload):
    """the newest committed SQ instruction pass of this workload's throughput-mode job (tools/pmc_sq_job.py) collected on these sources, or None"""
    if workload != "dense_4m":
        return None
    try:
        for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if "_pmc_sq_bench" in name and name.endswith(".json"):
                doc = json.load(open(os.path.join(ROOT, "profiles", name)))
                if doc.get("population") == "dense_4m_throughput_mode_cohort_launches" and _same_sources(doc):
                    return dict(doc, source=f"profiles/{name}")
    except (OSError, ValueError, KeyError):
        pass
    return None


def tail_roofline():
    """the one-workgroup protocol tails against their own floor — a member of a merged k_logup_tail launch is a chain of sponge permutations on ONE wave: floor =
    permutations per member x the permutation's cost in a loop (profiles/r04_p2l_bench_limb_sponge.txt) — from the diagnostic-build timing of this build
    (tools/tail_roofline.py -> profiles/r*_tail_roofline.json: entry -> exit of every member of every merged launch at 448 proofs in flight), or None"""
    try:
        for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if name.endswith("_tail_roofline.json"):
                doc = json.load(open(os.path.join(ROOT, "profiles", name)))
                if _same_sources(doc):
                    return dict(doc, source=f"profiles/{name}")
    except (OSError, ValueError, KeyError):
        pass
    return None


def pmc_traffic(kernel, population, launches=None):
    """HBM bytes per launch of `kernel` (mean over its launches) from a committed rocprofv3 PMC pass (profiles/r*_pmc_*.json, made by
    tools/pmc_summary.py: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled for 16 B/lane streaming reads as
    MI355X_MICROARCH.md prescribes) — ONLY from a file that names the same launch population (`population`: which command, which
    launches) and the exact instantiation, and, when `launches` is given, the same number of launches per unit. Anything else is None:
    round 2 quoted a figure collected on another build from a trace that included Context::generate."""
    try:
        norm = lambda k: k.replace(" ", "")  # noqa: E731
        for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if "_pmc_" not in name or not name.endswith(".json"):
                continue
            doc = json.load(open(os.path.join(ROOT, "profiles", name)))
            if doc.get("population") != population or not _same_sources(doc):  # (another population, or a pass collected on other sources than the ones that run)
                continue
            for rec in doc["kernels"]:
                if "hbm_bytes_per_launch" in rec and norm(rec["kernel"]) == norm(kernel):
                    per_unit = doc.get("units")
                    if launches is not None and per_unit and rec["launches"] != launches * per_unit:
                        return None
                    return dict(rec, source=f"profiles/{name}")
    except (OSError, ValueError, KeyError):
        pass
    return None


def spawn_ranks(n):
    """re-run this command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` (rendezvous on 127.0.0.1, a
    free port); rank 0's JSON line is the only thing the ranks print on stdout, so the child's stdout IS the bench line"""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc:
        sys.exit(rc)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dense_4m", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sumcheck24", action="store_true", help="skip the standalone 2^24 sumcheck roofline section")
    ap.add_argument("--no-cnn", action="store_true", help="skip the CNN-264k section of a Dense-4M run")
    ap.add_argument("--no-transformer", action="store_true", help="skip the transformer-layer section (models.transformer_layer: LayerNorm, QKV, the Mha node, feed-forward half)")
    ap.add_argument("--no-seam-level", action="store_true", help="skip the seam-level consumer section (tests/support/seam_bench.c)")
    ap.add_argument("--no-batch64", action="store_true", help="skip the config-4 anchor (one batch of 64 proofs per step on this GPU) of a default run")
    ap.add_argument("--batch", type=int, default=0, help="BASELINE config 4: ONE fixed batch of this many proofs per step, split over the ranks (strong scaling); "
                                                         "0 = the default weak-scaling run (fixed work per GPU)")
    ap.add_argument("--concurrency", type=int, default=0, help=f"independent proofs in flight per GPU (0 = {DEFAULT_IN_FLIGHT})")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher — one rank per GPU under torch.distributed.run
        # (the driver's multi-GPU command line does this itself; a plain invocation used to die on the WORLD_SIZE assert below)
        return spawn_ranks(args.gpus)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # one hardware queue per in-flight proof stream, as many as the GPU serves without time slicing
    if os.environ.get("DP_BENCH_NO_TORCH") and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # quick single-GPU checks on a fresh box, where the first `import torch` alone costs 1-2 minutes: no barrier is needed
        # at N=1 and every prove_batch returns only when its proofs are complete and downloaded, so the device is idle at both
        # ends of the timed region without torch.cuda.synchronize()
        class torch:  # noqa: N801
            class cuda:  # noqa: N801
                is_available = staticmethod(lambda: False)
                set_device = staticmethod(lambda d: None)
    else:
        import torch
    import numpy as np  # noqa: F401
    import deep_prove_amd as dpa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("DP_FORCE_DEVICE", local_rank)))
        # DP_DIST_BACKEND / DP_FORCE_DEVICE: let several ranks share one GPU over gloo (validating the N>1 path on a 1-GPU box)
        dist.init_process_group(backend=os.environ.get("DP_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("DP_BENCH_LAUNCH_CHECK"):
        # launch-path check (tests/test_distributed.py, GPU-less box): the ranks exist, found each other, and agree on the world
        te = torch.ones(1, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(te)
        if rank == 0:
            print(json.dumps({"launch_check": True, "world": world, "ranks_seen": int(te.item()), "batch": args.batch}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return None
    if "DP_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["DP_FORCE_DEVICE"])
        os.environ.setdefault("DP_HBM_FRACTION", f"{0.8 / max(1, local_world):.3f}")  # the ranks share ONE GPU's memory: each sizes its workers against its share
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    # host threads per rank: this rank's share of the CPUs the job may use (cgroup quota), two left to the HIP runtime
    budget = dpa.api.host_cpu_budget()
    host_threads = max(1, int(budget / max(1, local_world)) - 2)
    if local_world > 1:  # (one rank on the node: the library's own default — one idle-sleeping thread per cohort — applies; several ranks share the CPUs: each gets its share)
        os.environ.setdefault("DP_HOST_THREADS", str(host_threads))
    host_threads = int(os.environ.get("DP_HOST_THREADS", host_threads))
    # A rank needs ~4 ms of host work per Dense-4M proof (transcript, claims, launch packs: DESIGN.md §7): ONE thread sustains ~250 proofs/s, the GPU
    # ~500. With fewer than 3 proving threads per rank the host, not the GPU, bounds the rate — say so instead of letting it read as a scaling loss.
    host_bound = host_threads < MIN_HOST_THREADS_PER_RANK
    if host_bound and rank == 0:
        print(f"[bench] WARNING: {budget:.1f} usable CPUs for {local_world} rank(s) on this node = {host_threads} proving thread(s) per rank "
              f"(< {MIN_HOST_THREADS_PER_RANK}): this run is HOST-BOUND; give every GPU at least {MIN_HOST_THREADS_PER_RANK + 2} CPUs", file=sys.stderr, flush=True)
    conc = args.concurrency if args.concurrency > 0 else DEFAULT_IN_FLIGHT

    dev = dpa.Device(local_rank)
    main_w = measure_workload(dpa, dev, args.workload, conc, args.steps, args.warmup, world, rank, dist, torch, profile=rank == 0, strong_batch=args.batch)
    cnn_w = None
    if args.workload == "dense_4m" and not args.no_cnn and not args.batch:
        cnn_w = measure_workload(dpa, dev, "cnn_264k", conc, cnn_steps(args.steps), min(1, args.warmup), world, rank, dist, torch)

    per_rank = None
    if dist is not None and world > 1:  # what every rank saw: a slow rank (setup, one long step) must be visible in the N > 1 line
        mine = {"rank": rank, "step_ms": [round(v, 1) for v in main_w["step_ms"]], "setup_s": round(main_w["setup_s"], 2), "single_proof_latency_ms": round(main_w["latency_ms"], 2),
                "host_threads": host_threads, "in_flight": int(main_w["in_flight"])}
        per_rank = [None] * world
        try:
            dist.all_gather_object(per_rank, mine)
        except Exception as e:  # noqa: BLE001
            per_rank = [mine, {"error": f"all_gather_object: {type(e).__name__}: {e}"[:200]}]
    result = None
    if rank == 0:
        def rate(w, steps):
            return (steps * args.batch if args.batch else world * steps * BATCHES_PER_STEP * conc) / w["elapsed"]
        value = rate(main_w, args.steps)
        # ---- roofline. The proof is integer work with no dense contraction (MFMA unused by design). Its chip-filling work is
        # Poseidon2: every Merkle node is one compress() = 2 permutations = ~1040 Goldilocks multiplications for 96 B moved
        # (10.8 mul/B: VALU-integer bound, HBM irrelevant), so the roofline is priced in compress()/s against the rate the same
        # kernel reaches on a chip-filling layer, measured NOW on this GPU (dp_probe_compress_rate, HIP events on the launch
        # stream). `achieved`/`frac` follow the contract (per launch of the dominant chip-filling kernel, HIP-event timed in
        # one latency-mode proof: small layers cannot fill 256 CUs alone); `job_*` price the whole timed job, where the
        # layers of the proofs in flight share the chip.
        rep = main_w["kernel_report"]
        tot_ms = sum(r["total_ms"] for r in rep)
        alg_bytes_per_proof = sum(r["alg_bytes"] for r in rep)
        merkle = [r for r in rep if r["kernel"].startswith("k_merkle_layer")]
        nodes_per_proof = sum(r["alg_bytes"] for r in merkle) / 96.0
        peak = max(dev.probe_compress_rate(1 << 21, 8) for _ in range(4))  # best of 4 bursts: a single short burst right after a batch can catch the clocks ramping
        # the same probe held for ~1.5 s with the clock sampler on: what the kernel sustains once the power management has settled (the bursts above are 8 ms long)
        sustained = None
        try:
            smp = ClockSampler(int(os.environ.get("LOCAL_RANK", "0"))).start()
            t_s, rates = time.perf_counter(), []
            while time.perf_counter() - t_s < 1.5:
                rates.append(dev.probe_compress_rate(1 << 21, 32))
            sck = smp.stop()
            tail_rates = rates[len(rates) // 2:]  # the second half: after the clocks have settled
            sustained = {"compress_per_s": round(sum(tail_rates) / len(tail_rates) / 1e9, 4), "bursts_of_32_launches": len(rates), "first_burst": round(rates[0] / 1e9, 4),
                         "last_burst": round(rates[-1] / 1e9, 4), "gpu_clocks": sck}
        except Exception as e:  # noqa: BLE001
            sustained = {"error": f"{type(e).__name__}: {e}"}
        wide = [r for r in merkle if r["kernel"] == "k_merkle_layer"]  # the one-node-per-lane kernel of the wide layers (the _lp / tail variants serve layers too narrow to fill the chip)
        dom = wide[0] if wide else max(merkle, key=lambda r: r["total_ms"]) if merkle else rep[0]
        avg_ms = dom["total_ms"] / dom["launches"]
        nodes_per_launch = dom["alg_bytes"] / dom["launches"] / 96.0
        achieved = nodes_per_launch / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
        job_compress = nodes_per_proof * value / world  # per GPU
        pmc = pmc_traffic(dom["kernel"], "dense_4m_latency_proofs", dom["launches"])
        by_time = rep[0]
        roofline = {"bound": "valu-int", "kernel": dom["kernel"], "achieved": round(achieved / 1e9, 4), "peak": round(peak / 1e9, 4), "unit": "Gcompress/s",
                    "frac": round(achieved / peak, 4) if peak else None, "traffic": pmc["hbm_bytes_per_launch"] if pmc else None, "traffic_source": pmc["source"] if pmc else None,
                    "alg_bytes_per_launch": round(dom["alg_bytes"] / dom["launches"], 1), "nodes_per_launch": round(nodes_per_launch, 1),
                    "launches_per_proof": dom["launches"], "avg_launch_us": round(1000 * avg_ms, 3),
                    "peak_note": "k_merkle_layer on a 2^21-node layer, best of 4 bursts of 8 launches, HIP events, this run; 1 compress = 2 Poseidon2-w8 permutations = ~1040 Goldilocks multiplications",
                    "peak_sustained": sustained, "job_frac_of_sustained_peak": round(job_compress / (1e9 * sustained["compress_per_s"]), 4) if sustained and sustained.get("compress_per_s") else None,
                    "job_compress_per_s": round(job_compress / 1e9, 4), "job_frac": round(job_compress / peak, 4) if peak else None,
                    "job_goldilocks_mul_per_s": round(1040.0 * job_compress / 1e12, 4), "merkle_nodes_per_proof": int(nodes_per_proof),
                    "job_alg_GBps": round(alg_bytes_per_proof * value / world / 1e9, 1), "job_hbm_frac": round(alg_bytes_per_proof * value / world / 1e9 / HBM_PEAK_GBS, 5),
                    "alg_bytes_per_proof": int(alg_bytes_per_proof), "gpu_busy_ms_per_proof_latency_mode": round(tot_ms, 3),
                    "largest_by_time": {"kernel": by_time["kernel"], "bound": "latency", "share_of_gpu_time": round(by_time["total_ms"] / tot_ms, 4), "launches": by_time["launches"],
                                        "note": "one-workgroup protocol kernels (sumcheck rounds + Fiat-Shamir) are latency-bound by construction: kilobytes of tables, a dependent chain of rounds"},
                    "top_kernels": [{"kernel": r["kernel"], "launches": r["launches"], "total_ms": round(r["total_ms"], 3),
                                     "GBps": round((r["alg_bytes"] / max(r["total_ms"], 1e-9)) / 1e6, 1)} for r in rep[:8]]}
        # ---- the hardware-derived VALU bound next to the self-probed peak (round 4): a counter pass over the same command (tools/pmc_sq_job.py ->
        # profiles/r*_pmc_sq_bench448.json: SQ_INSTS_VALU of the cohort launches per proof, and of the compress probe per compress) prices every VALU wave
        # instruction at 4 cycles on 1024 SIMDs at 2.4 GHz. `peak_valu_bound` = compress()/s if EVERY issue slot of the chip ran this kernel's instructions;
        # `valu_issue_util` = the share of the chip's issue slots the whole timed job uses at the rate measured NOW.
        sq = valu_accounting(args.workload)
        if sq:
            roofline["peak_valu_bound"] = round(sq["compress"]["peak_valu_bound_compress_per_s"] / 1e9, 4) if sq.get("compress") else None
            roofline["frac_of_valu_bound"] = round(achieved / sq["compress"]["peak_valu_bound_compress_per_s"], 4) if sq.get("compress") else None
            roofline["probe_frac_of_valu_bound"] = round(peak / sq["compress"]["peak_valu_bound_compress_per_s"], 4) if sq.get("compress") else None
            roofline["valu_instr_per_compress"] = sq["compress"]["valu_instr_per_compress"] if sq.get("compress") else None
            roofline["valu_wave_instr_per_proof"] = sq["valu_wave_instr_per_proof"]
            roofline["valu_issue_util"] = round(sq["valu_wave_instr_per_proof"] * (value / world) / sq["chip_issue_capacity_wave_instr_per_s"], 4)
            roofline["valu_hash_share"] = sq.get("hash_share"); roofline["valu_one_wave_share"] = sq.get("one_wave_share")
            # the same share priced at the shader clock the GPU actually held during the timed steps (ClockSampler) instead of the 2.4 GHz the counter pass assumes
            ck = (main_w.get("clocks") or {}).get("sclk_mhz")
            if ck and ck.get("mean") and sq.get("assumed", {}).get("clock_hz"):
                roofline["sclk_mhz_timed_region"] = ck["mean"]
                roofline["valu_issue_util_at_sampled_clock"] = round(roofline["valu_issue_util"] * sq["assumed"]["clock_hz"] / (1e6 * ck["mean"]), 4)
            roofline["valu_source"] = sq["source"]
        cpu = None if (world > 1 or args.no_cpu_baseline) else cpu_baseline(main_w["mb"], args.workload)
        sc24 = None if (world > 1 or args.no_sumcheck24) else sumcheck24(dev, dpa)
        cnn = None
        if cnn_w is not None:
            csteps = cnn_steps(args.steps)
            cnn = {"metric": "proofs/sec (prover), CNN-264k", "value": round(rate(cnn_w, csteps), 4), "unit": "proofs/s",
                   "ms_per_step": round(1000.0 * cnn_w["elapsed"] / csteps, 3), "steps": csteps, "proofs_per_step_per_gpu": BATCHES_PER_STEP * conc, "proofs_in_flight_per_gpu": cnn_w["in_flight"],
                   "single_proof_latency_ms": round(cnn_w["latency_ms"], 2), "gpu_clocks_timed_region": cnn_w.get("clocks"), "vs_baseline": round(rate(cnn_w, csteps) / PUBLISHED["cnn_264k"], 3),
                   "baseline_note": "reference README.md:17 CNN-264k proving time 1242 ms on unstated CPU hardware",
                   "workload": WORKLOADS["cnn_264k"], "proof_words": cnn_w["proof_words"], "setup_s": round(cnn_w["setup_s"], 2), "verified": True, "golden_sha256_ok": cnn_w["golden_ok"], "verified_proofs_of_last_step": cnn_w["verified"], "verify_batch_ms_per_proof": cnn_w["verify_batch_ms_per_proof"],
                   "cpu_baseline": None if (world > 1 or args.no_cpu_baseline) else cpu_baseline(cnn_w["mb"], "cnn_264k")}
        result = {
            "metric": {"dense_4m": "proofs/sec (prover), Dense-4M", "cnn_264k": "proofs/sec (prover), CNN-264k"}.get(args.workload, "proofs/sec (prover), MLP-w256"),
            "value": round(value, 4), "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * main_w["elapsed"] / args.steps, 3), "step_ms_min_median_max": ([min(main_w["step_ms"]), sorted(main_w["step_ms"])[len(main_w["step_ms"]) // 2], max(main_w["step_ms"])] if main_w["step_ms"] else None), "higher_is_better": True, "scaling": "strong" if args.batch else "weak",
            "vs_baseline": round(value / PUBLISHED[args.workload], 3) if args.workload in PUBLISHED else None,
            "baseline_note": "reference README.md:17-18 proving times (Dense-4M 2335 ms, CNN-264k 1242 ms) on unstated CPU hardware",
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload], "arithmetic": "Goldilocks p = 2^64 - 2^32 + 1 and its degree-2 extension (canonical u64 words)",
                       "proofs_per_step_per_gpu": (strong_share(args.batch, world, 0) if args.batch else BATCHES_PER_STEP * conc), "proofs_in_flight_per_gpu": main_w["in_flight"],
                       "proofs_per_step_all_gpus": (args.batch if args.batch else world * BATCHES_PER_STEP * conc),
                       "strong_scaling_note": (f"BASELINE config 4: one batch of {args.batch} independent proofs per step split over {world} GPU(s) (rank r proves proofs r, r+{world}, ...); "
                                               "every rank commits the model itself (Context::generate recomputed per rank, outside the timed region), no data-path collective") if args.batch else None,
                       "single_proof_latency_ms": round(main_w["latency_ms"], 2), "single_proof_latency_samples_ms": main_w["latency_samples_ms"], "first_proof_ms": round(main_w["first_ms"], 2),
                       "host_cpu_budget": budget, "host_threads_per_rank": host_threads, "host_bound": bool(host_bound),
                       "gpu_clocks_timed_region": main_w.get("clocks"),
                       "min_cpus_per_gpu": MIN_HOST_THREADS_PER_RANK + 2, "per_rank": per_rank,
                       "parallelism": f"replicas x{world} GPUs x {main_w['in_flight']} proofs in flight per GPU in lock-step cohorts of {os.environ.get('DP_COHORT') or -(-main_w['in_flight'] // 22)} (independent proofs, no data-path collective)",
                       "proof_words": main_w["proof_words"], "setup_s": round(main_w["setup_s"], 2), "verified": True, "golden_sha256_ok": main_w["golden_ok"], "verified_proofs_of_last_step": main_w["verified"], "verify_ms_per_proof": main_w["verify_ms"], "verify_batch_ms_per_proof": main_w["verify_batch_ms_per_proof"],
                       "env_knobs": {k: v for k, v in sorted(os.environ.items()) if k.startswith("DP_") or k == "GPU_MAX_HW_QUEUES"}, "device": dev.name},
            "roofline": roofline, "cpu_baseline": cpu, "cnn_264k": cnn, "sumcheck24": sc24, "sumcheck24_sharded": None,
            "seam_level": None if (world > 1 or args.no_seam_level) else seam_level(host_threads, batch_rate=value),
        }
        result["tail_roofline"] = tail_roofline()
        if sc24 is not None and not args.batch:
            # the size sumcheck/benches/devirgo_sumcheck.rs itself runs, on ONE GPU: the anchor of the sharded 2^26 run (`sumcheck26_sharded` of an N > 1 line; the model of
            # DESIGN.md §7 says 2^24 on 8 GPUs is SLOWER than on one — 24 exchanges of ~35 us against 0.3 ms of streaming — and that 2^26 is where sharding starts to pay).
            # No oracle golden at this size (the oracle needs ~80 s of one core): the proof is checked by the host verifier and by its final evaluations.
            try:
                result["sumcheck26"] = sumcheck24(dev, dpa, nv=26)
            except Exception as e:  # noqa: BLE001
                result["sumcheck26"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and args.workload == "dense_4m" and not args.batch and not args.no_batch64:
            # BASELINE config 4 at N = 1: ONE batch of 64 independent proofs per step, all in flight at once — the anchor the 8-GPU strong-scaling line
            # (`bench.py --gpus 8 --batch 64`) is read against; the golden input rides in the last step and its proof must have the oracle's sha256
            try:
                b_steps = 6  # (six 0.14 s steps: one step of three now and then takes 220 ms instead of 137 — the first batch after the large job's workers were torn down)
                b64 = measure_workload(dpa, dev, "dense_4m", conc, b_steps, 1, world, rank, dist, torch, strong_batch=64)
                result["batch64"] = {"metric": "proofs/sec (prover), Dense-4M, one batch of 64 per step (BASELINE config 4 on 1 GPU)", "value": round(b_steps * 64 / b64["elapsed"], 4), "unit": "proofs/s",
                                     "ms_per_batch": round(1000.0 * b64["elapsed"] / b_steps, 2), "steps": b_steps, "warmup": 1, "proofs_in_flight": b64["in_flight"], "scaling": "strong",
                                     "step_ms": [round(v, 1) for v in b64["step_ms"]], "ms_per_batch_median": round(sorted(b64["step_ms"])[len(b64["step_ms"]) // 2], 2), "golden_sha256_ok": b64["golden_ok"], "verified_proofs_of_last_step": b64["verified"],
                                     "command_for_n_gpus": "python bench.py --gpus N --batch 64"}
            except Exception as e:  # noqa: BLE001
                result["batch64"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and args.workload == "dense_4m" and not args.batch and not args.no_transformer:
            try:  # (a side section: whatever happens in it, the headline line above is printed)
                result["transformer_layer"] = transformer_layer_section(dpa, dev, conc=int(os.environ.get("DP_BENCH_TL_IN_FLIGHT", "320")))
            except Exception as e:  # noqa: BLE001
                result["transformer_layer"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    # BASELINE config 5 across the ranks: ONE 2^24 sumcheck, every rank owns a contiguous 1/N slice of each table and the
    # per-round shares are all-gathered over RCCL (deep_prove_amd/sharded.py). It runs AFTER the headline is complete and
    # under a watchdog: neither an exception nor a collective that never returns may take the JSON line down.
    if world > 1 and not args.no_sumcheck24:
        def give_up():
            if rank == 0:
                result["sumcheck24_sharded"] = {"error": f"no result within {SHARDED_WATCHDOG_S:.0f} s (watchdog)"}
                print(json.dumps(result), flush=True)
            else:
                time.sleep(5.0)  # let rank 0 print first
            os._exit(0)
        def both():
            # 2^24 (BASELINE config 5) and 2^26 (the size sumcheck/benches/devirgo_sumcheck.rs itself runs; above the model's crossover,
            # sharded_estimate): the second one is checked against rank 0's own unsharded proof of the same tables (no oracle golden at 2^26)
            r24 = sumcheck24_sharded(dev, dpa, dist, world, rank)
            r26 = sumcheck24_sharded(dev, dpa, dist, world, rank, nv=26, compare_unsharded=True)
            return {"nv24": r24, "nv26": r26}
        sharded = guarded(both, SHARDED_WATCHDOG_S, give_up)
        if rank == 0:
            result["sumcheck24_sharded"] = sharded.get("nv24", sharded) if isinstance(sharded, dict) else sharded
            result["sumcheck26_sharded"] = sharded.get("nv26") if isinstance(sharded, dict) else None
    if rank == 0:
        print(json.dumps(result), flush=True)
    dev.close()
    if dist is not None:
        dist.destroy_process_group()
    return result


def sumcheck24(dev, dpa, nv=24, k=3):
    """BASELINE config 5 on one GPU: standalone sumcheck of one product of k base-field MLEs with 2^nv entries
    (sumcheck/benches/devirgo_sumcheck.rs shape). The timed proof is checked: its sha256 must equal the oracle's committed one
    (tests/golden/sumcheck24.json) and the host verifier must accept it. HBM roofline of the fused fold+sum kernel from HIP-event
    timings: MEDIAN over 5 profiled repetitions after one profiled warm-up (round 2 took one cold repetition and reported half
    the rate profiles/ shows); no roofline is printed when the profiled kernel total exceeds 1.2 x the un-profiled wall."""
    import hashlib
    import numpy as np
    n = 1 << nv
    tabs = [dpa.Mle.from_base(dev, dpa.models.splitmix64(0xD33B0000 ^ (5 << 32) ^ j, n) % np.uint64(dpa.P)) for j in range(k)]
    vp = dpa.VirtualPolynomial(nv)
    vp.add_mle_list(tabs)
    dpa.prove_parallel(dev, vp, dpa.Transcript(b"test"))  # warm
    walls, proof, finals, tr = [], None, None, None
    for _ in range(5):
        tr = dpa.Transcript(b"test")
        t0 = time.perf_counter()
        proof, finals = dpa.prove_parallel(dev, vp, tr)
        walls.append(1000 * (time.perf_counter() - t0))
    wall_ms = sorted(walls)[len(walls) // 2]
    # ---- parity of what was timed (the last timed repetition)
    gold_all = json.load(open(os.path.join(ROOT, "tests", "golden", "sumcheck24.json")))
    gold = gold_all["cases"].get(str(nv)) if k == gold_all["k"] else None
    golden_ok = None
    if gold is not None:
        golden_ok = bool(proof.size == gold["proof_words"] and hashlib.sha256(proof.tobytes()).hexdigest() == gold["sha256"]
                         and [int(v) for v in finals] == gold["finals"] and list(tr.read_challenge()) == gold["next_challenge"])
        assert golden_ok, f"2^{nv} sumcheck: the timed proof differs from the oracle's proof stream"
    off = 1 + 2 * nv + 1 + 1  # [nv, point, rounds, len(msg 0)] then the first round message
    Pm = int(dpa.P)
    claimed = ((int(proof[off]) + int(proof[off + 2])) % Pm, (int(proof[off + 1]) + int(proof[off + 3])) % Pm)
    _, expected = dpa.verify_sumcheck(claimed, proof, nv, k, dpa.Transcript(b"test"))  # raises on rejection
    prod = (1, 0)
    for i in range(k):
        f = (int(finals[2 * i]), int(finals[2 * i + 1]))
        prod = ((prod[0] * f[0] + 7 * prod[1] * f[1]) % Pm, (prod[0] * f[1] + prod[1] * f[0]) % Pm)
    assert prod == tuple(expected), "sumcheck24: the final evaluations do not multiply to the verifier's sub-claim"
    # ---- per-kernel HIP-event timing: one profiled warm-up, then the median of 5 profiled repetitions
    dev.profile(True)
    dpa.prove_parallel(dev, vp, dpa.Transcript(b"test"))
    reps = []
    for _ in range(5):
        dev.profile(True)  # drops the records of the repetition before, keeps its events for reuse
        dpa.prove_parallel(dev, vp, dpa.Transcript(b"test"))
        reps.append(dev.profile_report())
    dev.profile(False)
    for t in tabs:
        t.free()

    def is_stream(r):
        return r["kernel"].startswith("k_sc_fused") or r["kernel"].startswith("k_sc_terms")
    med = lambda xs: sorted(xs)[len(xs) // 2]  # noqa: E731
    names = sorted({r["kernel"] for rep in reps for r in rep})
    per = {}
    for name in names:
        rows = [r for rep in reps for r in rep if r["kernel"] == name]
        if len(rows) == len(reps):  # present in every repetition with the same launch count
            per[name] = {"kernel": name, "launches": rows[0]["launches"], "alg_bytes": rows[0]["alg_bytes"], "total_ms": med([r["total_ms"] for r in rows])}
    stream = [r for r in per.values() if is_stream(r)]
    ms = sum(r["total_ms"] for r in stream)
    by = sum(r["alg_bytes"] for r in stream)
    # kernel records only: the staging copies of the profiled path ("memcpy_*" records bracket a copy AND the host wait behind it)
    # are not kernel time
    is_kernel = lambda r: r["kernel"].startswith("k_")  # noqa: E731
    prof_total_ms = med([sum(r["total_ms"] for r in rep if is_kernel(r)) for rep in reps])
    prof_other_ms = med([sum(r["total_ms"] for r in rep if not is_kernel(r)) for rep in reps])
    mid = sorted(reps, key=lambda rep: sum(r["total_ms"] for r in rep))[len(reps) // 2]
    # the dominant launch: the instantiation with the longest AVERAGE launch (the first fused fold+sum round over 2^24 entries).
    # The later rounds reuse one instantiation from 2^23 down to 2^13 entries, where a launch is latency- not bandwidth-sized:
    # their aggregate is `achieved_GBps_all_streaming_rounds`, every instantiation is in `kernels`
    big = max(stream, key=lambda r: r["total_ms"] / r["launches"])
    big_gbs = (big["alg_bytes"] / big["launches"]) / (big["total_ms"] / big["launches"] * 1e-3) / 1e9
    pmc = pmc_traffic(big["kernel"], f"sumcheck{nv}")  # (the counter pass of THIS size: round 5 quoted the 2^24 pass for the 2^26 object)
    trusted = prof_total_ms <= 1.2 * wall_ms
    roofline = {"bound": "hbm", "kernel": big["kernel"], "achieved": round(big_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(big_gbs / HBM_PEAK_GBS, 4), "traffic": pmc["hbm_bytes_per_launch"] if pmc else None, "traffic_source": pmc["source"] if pmc else None,
                "alg_bytes_per_launch": round(big["alg_bytes"] / big["launches"], 1),
                "avg_launch_us": round(1000 * big["total_ms"] / big["launches"], 2), "launches": big["launches"],
                "timing": "HIP events on the launch stream, median of 5 profiled repetitions after 1 profiled warm-up",
                "note": ("k_sc_fused2 (round 6, the two-round grid): both folds of rounds 1 and 2 and the sums of round 3 in ONE pass over the base tables — per 8 base entries of "
                         "each table (64 B read, 32 B written) 16 multiplications for the folds and 8 for the three extension products; k_sc_terms2 before it reads the tables once for "
                         "the 4 x 4 grid that answers rounds 1 and 2 (32 multiplications per 96 B: VALU-bound). DP_SC_GRID2=0: the round-by-round form (k_sc_terms + one k_sc_fused per round)")
                        if big["kernel"].startswith("k_sc_fused2") else
                        ("k_sc_terms2 (round 6, the two-round grid): ONE pass over the base tables answers rounds 1 and 2 — the 4 x 4 grid of sums over every quad's bilinear extension, "
                         "taken at the points {0, 1, oo, -1}^2: 32 Goldilocks multiplications and 36 modular subtractions per 96 B read: VALU-integer bound (~600 VALU instructions per quad, 768 resident workgroups), so its HBM "
                         "fraction is not its roofline; k_sc_fused2 after it (both folds + round 3, 604 MB) runs at ~4.4 TB/s. DP_SC_GRID2=0: the round-by-round form")
                        if big["kernel"].startswith("k_sc_terms2") else
                        ("a fold+sum pass does ~36 Goldilocks multiplications per 192 B moved; at the measured ~1.0e12 mul/s of the chip the "
                         "VALU-integer bound (~0.15 ms for the first fused round) is above the HBM bound (0.13 ms at 6.3 TB/s)")}
    return {"workload": f"standalone sumcheck, one product of {k} base MLEs, 2^{nv} entries each (BASELINE config 5 on 1 GPU)",
            "wall_ms": round(wall_ms, 3), "wall_ms_samples": [round(w, 3) for w in walls], "rounds": nv, "golden_sha256_ok": golden_ok, "verified": True,
            "streaming_kernels_ms": round(ms, 3), "profiled_kernel_total_ms": round(prof_total_ms, 3), "profiled_non_kernel_records_ms": round(prof_other_ms, 3),
            "profiled_records_median_repetition": [[r["kernel"], r["launches"], round(r["total_ms"], 4)] for r in mid], "alg_bytes": by,
            "alg_bytes_formula_48kN": 48 * k * n, "achieved_GBps_all_streaming_rounds": round(by / (ms * 1e-3) / 1e9, 1),
            "end_to_end_GBps": round(48 * k * n / (wall_ms * 1e-3) / 1e9, 1), "end_to_end_hbm_frac": round(48 * k * n / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": roofline if trusted else None,
            "roofline_withheld": None if trusted else f"profiled kernel total {prof_total_ms:.3f} ms exceeds 1.2 x the un-profiled wall {wall_ms:.3f} ms: the profiled run is not representative",
            "kernels": [{"kernel": r["kernel"], "launches": r["launches"], "total_ms": round(r["total_ms"], 4),
                         "GBps": round(r["alg_bytes"] / max(r["total_ms"], 1e-9) / 1e6, 1)} for r in sorted(per.values(), key=lambda r: -r["total_ms"])[:6]]}


def sumcheck24_sharded(dev, dpa, dist, world, rank, nv=24, k=3, compare_unsharded=False):
    """one 2^nv sumcheck over `world` GPUs: rank g holds entries [g N/W, (g+1) N/W) of each of the k base tables"""
    import numpy as np
    kk = world.bit_length() - 1
    if 1 << kk != world:
        return {"skipped": f"world size {world} is not a power of two"}
    n = 1 << nv
    chunk = n // world
    terms = [((1, 0), list(range(k)))]
    ex = dpa.sharded.TorchExchange()
    # the round loop runs IN the library over its own RCCL communicator (dp_sumcheck_prove_sharded: local round sums,
    # ncclAllGather of the shares on device buffers, mod-p sum + sponge on the host, no Python per round) when the job runs on real
    # GPUs; several ranks sharing one GPU over gloo (CPU validation of this path) keep the Python loop over torch.distributed
    group = dpa.sharded.RcclGroup(dev) if dist.get_backend() == "nccl" else None

    def once():
        # the slice of a SplitMix64 stream is the stream started `rank * chunk` steps later
        tabs = [dpa.Mle.from_base(dev, dpa.models.splitmix64(((0xD33B0000 ^ (5 << 32) ^ j) + rank * chunk * 0x9E3779B97F4A7C15) % (1 << 64), chunk) % np.uint64(dpa.P)) for j in range(k)]
        small = []

        def make_small(tw):
            ms = [dpa.Mle.from_ext(dev, w) for w in tw]
            small.extend(ms)
            return dpa.sharded.HipShard(dev, kk, ms, terms)
        if group is not None:
            dist.barrier()
            t0 = time.perf_counter()
            proof, finals = dpa.sharded.prove_sharded_in_library(dev, group, nv, tabs, terms, dpa.Transcript(b"test"))
            dt = time.perf_counter() - t0
        else:
            shard = dpa.sharded.HipShard(dev, nv - kk, tabs, terms)
            dist.barrier()
            t0 = time.perf_counter()
            proof, finals = dpa.sharded.prove_sharded([shard], ex, nv, terms, dpa.Transcript(b"test"), make_small)
            dt = time.perf_counter() - t0
        for m in tabs + small:
            m.free()
        return proof, dt
    once()
    times, proof = [], None
    for _ in range(3):
        proof, dt = once()
        times.append(dt)
    import torch
    te = torch.tensor([min(times)], dtype=torch.float64, device=ex.device)
    dist.all_reduce(te, op=dist.ReduceOp.MAX)
    import hashlib
    est = sharded_estimate(nv, k, world)
    sha = hashlib.sha256(proof.tobytes()).hexdigest()
    check = None
    gold_all = json.load(open(os.path.join(ROOT, "tests", "golden", "sumcheck24.json")))
    if k == gold_all["k"] and str(nv) in gold_all["cases"]:
        check = {"against": "oracle golden (tests/golden/sumcheck24.json)", "ok": sha == gold_all["cases"][str(nv)]["sha256"]}
    elif compare_unsharded and rank == 0:
        full = [dpa.Mle.from_base(dev, dpa.models.splitmix64(0xD33B0000 ^ (5 << 32) ^ j, n) % np.uint64(dpa.P)) for j in range(k)]
        vp = dpa.VirtualPolynomial(nv)
        vp.add_mle_list(full)
        t0 = time.perf_counter()
        solo, _ = dpa.prove_parallel(dev, vp, dpa.Transcript(b"test"))
        solo_ms = 1000 * (time.perf_counter() - t0)
        for m in full:
            m.free()
        check = {"against": "rank 0's unsharded prove_parallel of the same tables", "ok": hashlib.sha256(solo.tobytes()).hexdigest() == sha, "unsharded_wall_ms_cold": round(solo_ms, 3)}
    if check is not None:
        assert check["ok"], f"sharded 2^{nv} sumcheck: proof differs ({check['against']})"
    return {"a_priori_estimate": est, "parity": check, "workload": f"ONE standalone sumcheck, product of {k} base MLEs of 2^{nv} entries, sharded over {world} GPUs (contiguous slices, shares all-gathered per round)",
            "wall_ms": round(1000 * float(te.item()), 3), "rounds": nv, "local_rounds": nv - kk, "alg_bytes_per_gpu": 48 * k * chunk,
            "proof_sha256": sha, "round_loop": "in-library C++ over RCCL (dp_sumcheck_prove_sharded)" if group is not None else "Python over torch.distributed",
            "note": "the proof stream is bit-identical to the single-GPU prove_parallel of the same tables (same sha256 at every world size)"}


def sharded_estimate(nv, k, world, stream_tbps=3.4, round_us=17.0, exchange_us=35.0):
    """A-PRIORI model of the sharded sumcheck (an estimate printed next to the measurement, NOT a measurement): one GPU pays the
    streaming of 48 k N bytes at the rate the fused rounds reach (3.4 TB/s over all streaming rounds, profiles/r03_call1_bench.json)
    plus ~17 us per round of reduction + host round trip; W GPUs stream 1/W of it but pay an exchange per local round on top —
    ncclAllGather of <= 256 B over xGMI + the share-sum kernel + its publication, ~35 us assumed with the shares kept on the device
    (csrc/sharded.h; ~3 host hops more without). Sharding pays when (32 - 48/W) kN/BW > (nv - log2 W) exchange_us - round_us (round 6: the single GPU streams 32kN, the
    two-round grid)."""
    lg = world.bit_length() - 1
    n = 1 << nv
    stream_ms = 48.0 * k * n / (stream_tbps * 1e12) * 1e3
    # (round 6: ONE GPU runs the two-round grid — 32 k N bytes, one host round trip less; the shards of a sharded run keep the round-by-round form, their round sums being shares)
    one = 32.0 * k * n / (stream_tbps * 1e12) * 1e3 + (nv - 1) * round_us * 1e-3
    many = stream_ms / world + (nv - lg) * (round_us + exchange_us) * 1e-3 + lg * round_us * 1e-3
    cross = next((v for v in range(lg + 1, 40) if (32.0 - 48.0 / world) * k * (1 << v) / (stream_tbps * 1e12) * 1e6 > (v - lg) * exchange_us - round_us), None)
    return {"one_gpu_ms": round(one, 3), f"{world}_gpus_ms": round(many, 3), "assumed": {"streaming_TBps": stream_tbps, "round_us": round_us, "exchange_us_per_local_round": exchange_us},
            "crossover_nv": cross, "verdict": ("worth sharding" if many < one else f"not worth sharding at 2^{nv} on {world} GPUs: the per-round exchange outweighs the streaming saved (crossover 2^{cross})"),
            "note": "model, not measurement; the measured wall_ms of this section is the judge of it"}


def seam_level(threads, per_thread=6, batch_rate=None):
    """What a host gets that proves THROUGH THE SEAMS (dp_pcs_commit / dp_pcs_batch_open / dp_sumcheck_prove / dp_logup_prove from T
    threads with one context each) instead of handing the model to dp_model_prove_batch: tests/support/
    seam_bench.c replays the seam calls of one Dense-4M proof, call for call and shape for shape, on random tables — "workload-
    equivalent" proofs (the seams' bit-exactness is the business of the parity tests). Run in its own process (own HIP contexts)."""
    import subprocess
    import deep_prove_amd as dpa
    src = os.path.join(ROOT, "tests", "support", "seam_bench.c")
    out = os.path.join(ROOT, "tests", "support", "_build", "seam_bench")
    deps = [src, os.path.join(ROOT, "include", "deep_prove_hip.h"), dpa.LIB_PATH]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-O2", "-o", out, src, "-L", os.path.dirname(dpa.LIB_PATH), "-ldeepprove_hip", "-lpthread",
                               "-Wl,-rpath," + os.path.dirname(dpa.LIB_PATH)])
    res = {}
    # `streams_throughput`: plain contexts switched to throughput mode (dp_ctx_set_throughput_mode: device-side Fiat-Shamir, fused protocol
    # kernels). Measured: 40 against 80 proofs/s at 14 threads, 34-36 with 28 / 56 yielding threads (profiles/r03_seam_level_throughput_mode.txt):
    # without cohorts to merge launches and fibers to keep hundreds of calls in flight the one-wave kernels only add latency
    # `async_one_thread` (round 4): ONE host thread keeps 128 / 384 proofs in flight through the submit / poll forms (dp_async): calls of identical shape are
    # merged into lock-step groups by the engine (tests/support/seam_bench.c mode 3)
    # (192 in flight — six groups of 32 per call shape — is where the engine's grouping works best: 480-497 proofs/s against 373-432 at 128 / 160 / 224 / 256 / 384, tools/r06/call49.sh)
    # `blocking_routed_T` (round 6): T host threads making the BLOCKING seam calls on one context routed to one engine (dp_ctx_route_to_engine: every call is a
    # submit + wait, calls of one shape from different threads are merged; tests/support/seam_bench.c mode 4) — the form a host written against the reference's
    # synchronous traits has; its rate is bounded by T / (latency of one proof's chain of calls), so it is quoted at the thread count of `streams` and at 128
    variants = [("streams", 0, threads, {}), ("streams_throughput", 2, threads, {}), ("async_one_thread_128", 3, 128, {}), ("async_one_thread_192", 3, 192, {}), ("async_one_thread_384", 3, 384, {}),
                (f"blocking_routed_{threads}", 4, threads, {}), ("blocking_routed_128", 4, 128, {})]
    for name, executor, t, extra in variants:
        env = dict(os.environ, DP_ARENA_BYTES=str(2 << 30), **extra)
        try:
            r = subprocess.run([out, str(t), str(3 if executor == 3 or t > 64 else per_thread), str(executor)], env=env, capture_output=True, text=True, timeout=120)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            res[name] = json.loads(line[-1]) if r.returncode == 0 and line else {"error": (r.stderr or r.stdout)[-300:]}
        except Exception as e:  # noqa: BLE001
            res[name] = {"error": f"{type(e).__name__}: {e}"}
    res["note"] = ("workload-equivalent proofs per second of a seam-level host (every seam call of one Dense-4M proof, random tables), T threads with one dp_ctx each: "
                   "`streams` = plain contexts in latency mode (one HIP stream each), `streams_throughput` = the same in throughput mode (dp_ctx_set_throughput_mode), `async_one_thread_N` = one host thread, N proofs in flight through dp_async (submit / poll, "
                   "merged lock-step groups), `blocking_routed_T` = T threads making the blocking calls on one context routed to one engine (dp_ctx_route_to_engine: submit + wait, merged across threads); "
                   "`vs_batch` = best seam-level rate / the batch rate of this run")
    rates = [v.get("seam_level_proofs_per_s", 0.0) for v in res.values() if isinstance(v, dict)]
    res["best_proofs_per_s"] = max(rates) if rates else None
    res["vs_batch"] = round(max(rates) / batch_rate, 4) if rates and batch_rate else None
    return res


def cpu_baseline(mb, workload):
    """the oracle ("port" of the reference CPU path) on a bounded sample of the same workload: one proof on one core (the
    latency), then one proof per host core in parallel (the throughput leg: independent replicas, the CPU analogue of the
    proofs the GPU keeps in flight)"""
    import numpy as np
    import deep_prove_amd as dpa
    from support import oracle_lib
    o = oracle_lib.load()
    h = o.model_setup(mb.blob())
    x = mb.input(1000)
    proof, _, ms1 = o.model_prove(h, x)
    cores = max(1, int(dpa.api.host_cpu_budget()))
    wall, dg = o.model_prove_many(h, x, cores, 1)
    word_sum = int(proof.sum(dtype=np.uint64))
    assert dg == (word_sum * cores) % (1 << 64), "CPU replicas produced different proofs"
    # SURVEY 8(d)'s planned baseline: ONE proof on all cores, the O(n) loops chunked rayon-style (oracle/par.hpp, with_min_len(64)),
    # prove() only as zkml/src/bin/bench.rs:390-408 times it; must produce the very same proof stream
    mt_ms, mt_dg = o.model_prove_mt(h, x, cores)
    o.model_free(h)
    assert mt_dg == word_sum % (1 << 64), "the multi-threaded CPU proof differs from the single-threaded one"
    return {"value": round(cores * 1000.0 / wall, 5), "unit": "proofs/s", "cores": cores, "kind": "port",
            "single_core_ms": round(ms1, 1),
            "sample": f"{cores} independent proofs of the same {workload} model on {cores} host threads (one each), prove() only "
                      f"(setup and inference excluded as in the reference harness): {wall:.0f} ms wall; one proof alone on one core: {ms1:.0f} ms",
            "port_mt": {"kind": "port-mt", "cores": cores, "ms_per_proof": round(mt_ms, 1), "value": round(1000.0 / mt_ms, 5), "unit": "proofs/s",
                        "speedup_over_one_core": round(ms1 / mt_ms, 2),
                        "sample": f"ONE {workload} proof on {cores} threads: the oracle's O(n) loops chunked over a fork-join pool with rayon's with_min_len(64) "
                                  "(the reference's rayon path restated; same proof bytes as the single-threaded run, checked)"}}


if __name__ == "__main__":
    main()
