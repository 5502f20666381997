#!/usr/bin/env python3
"""bench.py -- spectrogram frames/sec of the CR-CED-16 (V3) forward pass on N MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  * N > 1 under torch.distributed.run (RANK set): one rank per GPU over RCCL.
  * N > 1 started bare (`python bench.py --gpus 8`, RANK unset): this process starts the N ranks as a CHILD
    (`python -m torch.distributed.run --nproc-per-node N ... bench.py ...`) BEFORE anything here touches torch or
    HIP, relays the child's one JSON line and exits with its code.
Utterances shard over the batch axis with NO data-path collective (each rank owns B utterances resident in
its HBM) -> weak scaling; that is `value`.  The reference's own calling convention -- one host process holds the
whole batch (model_utils/tester.py:85-90) -- is measured beside it as `from_root`: scatter over RCCL, compute,
gather (fullycnnspeechenhancement_amd/dist.py).
A "step" = one forward of the hot path (model_utils/model.py:93-96 via the C ABI) over one batch of
B x T x 129 synthetic magnitude frames already resident in HBM.
Prints ONE JSON line on rank 0 with `roofline` (dominant kernel vs the fp32 MFMA/VALU peak, HIP events inside
the timed region), `cpu_baseline` (the CPU restatement timed on this box's cores) and, at N = 1, `secondary`
(BASELINE configs 2, 5 and 1, R-CED V1 / V2 in fp32 at config 3's shape, and the PCM -> PCM pipeline, under the same clock).
"""

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3   # MI355X dense fp32, vector = matrix (MI355X_MICROARCH.md chip table)
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (same table); only for bf16 lines
# Which matrix pipe a kernel's arithmetic runs on, and that pipe's peak in fp32-equivalent TFLOP/s, all from the guide's chip
# table (MI355X_MICROARCH.md:41-43):
#   mfma_f32     v_mfma_f32_16x16x4_f32, the fp32 pipe: 157.3 (= the vector rate)
#   mfma_bf16x6  fp32-quality products as six v_mfma_f32_16x16x32_bf16 over three-part operands: 2500 / 6 = 416.7
#                (a register-only loop on this part MEASURES 386: profiles/r03_f32_on_bf16.txt, kept as `measured_ceiling`)
#   mfma_bf16    plain bf16 operands: 2500
# Every roofline entry of the line is priced on the pipes the kernel runs on: `peak` = total FLOP / (sum over pipes of
# FLOP_i / peak_i), the rate a kernel with this mix of pipes reaches when each pipe runs at its spec peak, and `frac` =
# achieved / peak = (sum FLOP_i / peak_i) / time.  For a kernel that is pure mfma_f32 that is SURVEY 8(d3)'s achieved / 157.3;
# `frac_fp32_peak` (achieved / 157.3, may exceed 1 for a kernel on the faster pipe) stays beside it as an extra key.
PIPE_CEILING_TFLOPS = {"mfma_f32": FP32_PEAK_TFLOPS, "mfma_bf16x6": BF16_PEAK_TFLOPS / 6.0, "mfma_bf16": BF16_PEAK_TFLOPS}
MEASURED_CEILING_TFLOPS = {"mfma_bf16x6": 386.0}
NET_WORK = {1: "FullyCNN", 2: "FullyCNNV2", 3: "FullyCNNV3"}


def pipe_roofline(flop_by_pipe, seconds):
    """flop_by_pipe: {pipe: nominal dense FLOP in `seconds`} -> the roofline fields every kernel entry of the line carries:
    achieved, peak (of this mix of pipes at the guide's spec peaks), frac = achieved / peak."""
    flop_by_pipe = {k: float(v) for k, v in flop_by_pipe.items() if v > 0}
    total = sum(flop_by_pipe.values())
    achieved = total / seconds / 1e12
    floor_s = sum(v / (PIPE_CEILING_TFLOPS[k] * 1e12) for k, v in flop_by_pipe.items())
    out = {"bound": "mfma", "pipe": next(iter(flop_by_pipe)) if len(flop_by_pipe) == 1 else "mixed",
           "pipe_mix": {k: v / total for k, v in flop_by_pipe.items()},
           "pipe_peaks_tflops": {k: PIPE_CEILING_TFLOPS[k] for k in flop_by_pipe},
           "achieved": achieved, "unit": "TFLOP/s", "peak": total / floor_s / 1e12, "frac": floor_s / seconds,
           "frac_fp32_peak": achieved / FP32_PEAK_TFLOPS}
    if any(k in MEASURED_CEILING_TFLOPS for k in flop_by_pipe):
        meas = sum(v / (MEASURED_CEILING_TFLOPS.get(k, PIPE_CEILING_TFLOPS[k]) * 1e12) for k, v in flop_by_pipe.items())
        out["frac_measured_ceiling"] = meas / seconds
    return out


def layer_flops(spec, variant):
    """Nominal dense FLOP per frame of every layer (2 * MAC * 129 bins)."""
    return [2 * spec.FEATURE_DIM * l.kh * l.kw * l.cin * l.cout for l in spec.layers(variant)]


def forward_flops_by_pipe(spec, variant, kernel, dtype="f32", v3_l2x6=3):
    """FLOP per frame of one forward kernel kind, split by the pipe each layer runs on."""
    fl = layer_flops(spec, variant)
    if kernel == "conv_layer_generic":
        return {"mfma_f32": sum(fl)}            # direct fp32 FMA on the vector ALU: same 157.3 ceiling
    if variant == 3:                            # one kernel, all 16 layers.  Option v3_l2x6: 3 (the product) = every layer in the six-product
        # form; 2 = all but the first layer (1 -> 18 over 8 x 9) and decode_final, which stay on the fp32 MFMA; 1 = the five 18 -> 30 layers only;
        # 0 = every layer on the fp32 MFMA
        form = int(v3_l2x6)
        if form == 3:
            return {"mfma_bf16x6": sum(fl)}
        on_x6 = {0: (), 1: ((18, 30),), 2: ((18, 30), (30, 8), (8, 18))}[form]
        x6 = sum(f for f, l in zip(fl, spec.layers(3)) if (l.cin, l.cout) in on_x6)
        return {"mfma_f32": sum(fl) - x6, "mfma_bf16x6": x6}
    if dtype == "bf16":                         # frame16_kernel: ONE launch, every layer (first and output layers included) on the plain bf16 MFMA
        return {"mfma_bf16": sum(fl)} if kernel == "rced_fused" else {}
    if kernel == "rced_final_gemm":             # R-CED's 1x129 output layer (fp32 mode): x6::final_gemm_x6_kernel
        return {"mfma_bf16x6": fl[-1]}
    return {"mfma_f32": sum(fl[:-1])}


def train_flops_by_pipe(spec):
    """CR-CED training step, 3 x forward FLOPs (forward, dgrad, wgrad), by pipe: on the bf16 pipe in the
    three-part form run the 18 -> 30 forward convolutions, the 8 -> 30 dgrad inside the 30 -> 8 fused backward kernel, 16 of
    every 18 pixel groups of the 18 -> 30 wgrad, and the output layer's forward and dgrad; everything else on the fp32 MFMA."""
    fl, ly = layer_flops(spec, 3), spec.layers(3)
    f1830 = sum(f for f, l in zip(fl, ly) if (l.cin, l.cout) == (18, 30))
    f308 = sum(f for f, l in zip(fl, ly) if (l.cin, l.cout) == (30, 8))
    x6 = f1830 + f308 + f1830 * 16.0 / 18.0 + 2 * fl[-1]
    return {"mfma_f32": 3 * sum(fl) - x6, "mfma_bf16x6": x6}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU (config 3: 256)")
    ap.add_argument("--frames", type=int, default=512, help="time frames per utterance (config 3: 512)")
    ap.add_argument("--variant", type=int, default=3, choices=(1, 2, 3))
    ap.add_argument("--path", default="auto", choices=("auto", "layerwise", "fused"))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not time the dominant kernel with HIP events")
    ap.add_argument("--dtype", default="f32", choices=("f32", "bf16"),
                    help="bf16: R-CED V1/V2 only (BASELINE config 2); never the default, never the headline")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config 2 / config 5 entries (N = 1 only)")
    ap.add_argument("--from-root-steps", type=int, default=5, help="timed forward_from_root calls at N > 1 (0 = skip)")
    ap.add_argument("--from-root-chunks", type=int, default=8, help="pipeline depth of forward_from_root")
    ap.add_argument("--from-root-fail-status", type=int, default=3,
                    help="exit status of every rank when forward_from_root stalls (the line, with from_root.error, is printed "
                         "first and `launch_ranks` relays it whatever the status; 0 = report the stall in the line only)")
    ap.add_argument("--from-root-timeout", type=int, default=150,
                    help="seconds after which a stalled forward_from_root is abandoned and the line printed without it")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child process.  Nothing in this
    process has imported torch or touched HIP at this point (a process that has must never exec / be replaced);
    the .so is built first so that the ranks find it."""
    import socket
    import __graft_entry__ as ge
    ge.build_hip()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:            # relay: the one JSON line goes to stdout, anything else to stderr
        if out.lstrip().startswith("{") and '"metric"' in out:
            line = out
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    sys.exit(rc if rc != 0 or line is not None else 1)


def cpu_baseline(variant, weights, frames_t, budget_s):
    """The oracle's torch-CPU fp32 restatement (kind "port": TF 1.14 itself cannot run here) on a
    bounded sample of the same workload: batches of 8 or 32 utterances x T frames until ~budget_s.  oneDNN does
    not always scale to every hardware thread, so a short probe over (batch, threads) picks the fastest pair first;
    the probe table is part of `sample`."""
    import torch
    from oracle import rced_np, torch_ref
    ref = torch_ref.TorchRef(NET_WORK[variant], weights)
    x32 = torch.from_numpy(rced_np.make_input(32, frames_t, seed=1234))
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    cands = sorted({default_threads} | {c for c in (8, 16, 32, 64, 128, 256) if c <= ncpu})
    probe, best = {}, (8, default_threads)
    for nb in (8, 32):
        x = x32[:nb]
        for c in cands:
            if nb == 32 and c < 16 and len(cands) > 2:
                continue              # a big batch on few threads only costs probe time
            torch.set_num_threads(c)
            ref(x[:1])
            t0 = time.perf_counter()
            ref(x)
            probe[(nb, c)] = x.shape[0] * x.shape[1] / (time.perf_counter() - t0)
            if probe[(nb, c)] > probe.get(best, 0):
                best = (nb, c)
    nb, threads = best
    x = x32[:nb]
    torch.set_num_threads(threads)
    ref(x[:1])  # warm-up (oneDNN primitive creation)
    done, t0 = 0, time.perf_counter()
    while True:
        ref(x)
        done += x.shape[0] * x.shape[1]
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    torch.set_num_threads(default_threads)
    table = ", ".join("b%d/t%d: %.0f" % (k[0], k[1], v) for k, v in sorted(probe.items()))
    return {"value": done / el, "unit": "frames/s", "cores": int(threads), "kind": "port",
            "sample": "torch-CPU fp32 restatement (oracle/torch_ref.py), %d frames = %d batches of [%d,%d,129,1] in %.1f s "
                      "on %d threads; probe (batch/threads: frames/s): %s; host has %d logical cpus"
                      % (done, done // (nb * frames_t), nb, frames_t, el, threads, table, ncpu)}


def pmc_record(suffix, kernel_hash):
    """The newest committed rocprofv3 PMC capture profiles/rNN_<suffix> that was taken on exactly the kernel code being
    run, or (None, None): every capture records the hash of the sources that define the kernels and their packed-weight /
    LDS layouts (__graft_entry__.forward_kernel_hash / train_kernel_hash); a stale capture is never attached."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)), reverse=True):
        try:
            with open(path) as fh:
                d = json.load(fh)
        except Exception:
            continue
        if d.get("kernel_hash") == kernel_hash:
            return d, "profiles/" + os.path.basename(path)
    return None, None


def pmc_traffic(ge, variant, batch, frames, kernel):
    """HBM bytes per launch of the dominant kernel (FETCH_SIZE + WRITE_SIZE, separate passes, corrected as
    MI355X_MICROARCH.md prescribes; collected by tools/profile.sh on this same command)."""
    d, src = pmc_record("pmc_traffic.json", ge.forward_kernel_hash())
    if not d:
        return None, "no PMC capture for this kernel build (every profiles/r*_pmc_traffic.json records another kernel_hash)"
    wl = d["workload"]
    key = {"rced_fused": "fused_v3_kernel", "rced_final_gemm": "final_gemm_kernel"}.get(kernel)
    if key in d and (wl["variant"], wl["batch"], wl["frames"]) == (variant, batch, frames):
        return 1024 * (d[key]["fetch_kib"] + d[key]["write_kib"]), "%s (kernel_hash %s)" % (src, d["kernel_hash"])
    return None, "PMC capture is for another workload"


def synthetic_magnitudes(shape, seed):
    """x = |N(0,1)| float32 from numpy.random.default_rng(seed) (magnitude spectrograms are non-negative)."""
    import numpy as np
    x = np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)
    return np.abs(x, out=x)


def forward_line(args, torch, model, spec, _lib, variant, dtype, B, T, steps, warmup, world, rank, profile, ctl="cuda"):
    """Time `steps` forwards of one resident batch; returns (elapsed_s, per-kind HIP-event times or None)."""
    import torch.distributed as dist
    x = torch.from_numpy(synthetic_magnitudes((B, T, spec.FEATURE_DIM, 1), 1234 + rank)).cuda()   # resident before the clock starts
    y = torch.empty_like(x)
    model.reserve(B, T)
    lib = _lib.load()
    stream = torch.cuda.current_stream()
    h, xp, yp = model._handle, x.data_ptr(), y.data_ptr()

    def step():
        _lib.check(lib.rced_forward(h, xp, yp, B, T, stream.cuda_stream))

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if profile:
        model.profile(True)     # HIP events around every kernel launch, on the launch stream
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=ctl, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    times = None
    if profile:
        kinds = {_lib.K_GENERIC: "conv_layer_generic", _lib.K_FUSED: "rced_fused", _lib.K_FINAL: "rced_final_gemm"}
        times = {kinds[k]: model.profile_query(k) for k in kinds}
        model.profile(False)
    return elapsed, times, (x, y)


def host_buffers_line(model, x, steps=5):
    """The boundary as the reference calls it (tester.py:85-90): pageable numpy in, numpy out, through
    rced_forward_host (chunked H2D / kernel / D2H on three streams).  PCIe-inclusive; reported beside `value`."""
    import numpy as np
    xh = x.cpu().numpy()
    model(xh)                      # device staging buffers, streams and events are set up on the first call
    t0 = time.perf_counter()
    for _ in range(steps):
        yh = model(xh)             # a new output ndarray per call, as sess.run returns one -- over recycled warm pages (model.HostOutputPool:
                                   # the previous result is dropped here, as in the reference's loop, so its pages come back)
    el = time.perf_counter() - t0
    pool, model._host_pool = model._host_pool, None
    t0 = time.perf_counter()
    for _ in range(steps):
        yc = model(xh)             # ... and with numpy.empty per call: what the page faults of a fresh 67 MB array cost
    el_cold = time.perf_counter() - t0
    model._host_pool = pool
    del yc
    yo = np.empty_like(xh)
    model(xh, out=yo)
    t0 = time.perf_counter()
    for _ in range(steps):
        model(xh, out=yo)          # the caller's output array reused: no page faults of a fresh 67 MB buffer
    el_out = time.perf_counter() - t0
    n, t = xh.shape[0], xh.shape[1]
    return {"value": n * t * steps / el, "unit": "frames/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
            "ms_per_step_reused_output": 1e3 * el_out / steps, "ms_per_step_fresh_pages": 1e3 * el_cold / steps,
            "same_result": bool(np.array_equal(yh, yo)),
            "bytes_each_way": int(xh.nbytes), "finite": bool(abs(float(yh.sum())) < float("inf")),
            "note": "numpy [N,T,129,1] in -> numpy out through rced_forward_host: H2D + kernel + D2H per call, "
                    "PCIe-inclusive (SURVEY 8(d2) 'with-H2D/D2H figure'); never `value`.  ms_per_step: the default -- a new output "
                    "ndarray object per call, as sess.run returns one, over recycled already-touched pages (never pages the caller "
                    "still holds); ms_per_step_fresh_pages: numpy.empty per call (the OS zero-fills 67 MB: ~5 ms of page faults); "
                    "ms_per_step_reused_output: model(x, out=buf)"}


def from_root_line(args, torch, dist, model, spec, world, rank, dev_index, B, T, ctl="cuda", transports=("rccl", "copy")):
    """The reference's single-host-process convention: rank 0 holds the whole batch, every rank computes its batch slice, the masks
    end up on rank 0 (fullycnnspeechenhancement_amd/dist.py) -- over RCCL send / recv (transport "rccl", with r CUs left to the
    communicators' kernels, r swept) and as peer copies through IPC handles (transport "copy": no CU needed)."""
    from fullycnnspeechenhancement_amd.dist import BatchShardedForward, reserved_cus

    def timed(eng, steps, warm, **kw):
        for _ in range(warm):
            eng.forward_from_root(xr, root=0, chunks=args.from_root_chunks, **kw)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.forward_from_root(xr, root=0, chunks=args.from_root_chunks, **kw)
        torch.cuda.synchronize()
        dist.barrier()
        tm = torch.tensor([time.perf_counter() - t0], device=ctl, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        return float(tm.item())

    xr = None
    if rank == 0:
        xr = torch.from_numpy(synthetic_magnitudes((world * B, T, spec.FEATURE_DIM, 1), 1234)).cuda()
    steps = args.from_root_steps
    res = {}
    if "rccl" in transports:
        eng = BatchShardedForward(model, device="cuda:%d" % dev_index, forward_into=lambda a, out: model(a, out=out))
        # The fused kernel is a persistent grid of one workgroup per CU holding nearly all of the CU's LDS; RCCL's send / recv are
        # kernels too and can only start on a CU a workgroup has left.  So the pipelined call is timed with r CUs left free for
        # them (dist.reserved_cus: fused_grid = num_cus - r), r swept: every point is reported, `value` is the r = 0 figure and
        # `best` the fastest point of the sweep (a best-of-4: labelled as such).
        sweep = {}
        for r in (0, 4, 8, 16):
            with reserved_cus(model, r):
                sweep[r] = timed(eng, steps, 2 if r == 0 else 1)
        best_r = min(sweep, key=lambda k: sweep[k])
        # compute-free passes: what the links alone take per direction, so that the overlap shows as
        # ms_per_step ~ max(compute, transfer) instead of being inferred
        probe = {}
        for direction in ("scatter", "gather"):
            try:
                probe[direction + "_only_ms"] = 1e3 * timed(eng, 3, 1, direction=direction) / 3
            except Exception as e:
                probe[direction + "_only_ms"] = "%s: %s" % (type(e).__name__, e)
        eng.close()
        res["rccl"] = {"value": world * B * T * steps / sweep[0], "ms_per_step": 1e3 * sweep[0] / steps,
                       "reserved_cus_sweep_ms_per_step": {str(r): 1e3 * v / steps for r, v in sweep.items()},
                       "best": {"reserved_cus": best_r, "ms_per_step": 1e3 * sweep[best_r] / steps,
                                "value": world * B * T * steps / sweep[best_r], "note": "the fastest of the four sweep points"},
                       "transfer_only": probe}
    if "copy" in transports:
        try:
            eng = BatchShardedForward(model, device="cuda:%d" % dev_index, forward_into=lambda a, out: model(a, out=out), transport="copy")
            el = timed(eng, steps, 2)
            probe = {}
            for direction in ("scatter", "gather"):
                probe[direction + "_only_ms"] = 1e3 * timed(eng, 3, 1, direction=direction) / 3
            eng.close()
            res["copy"] = {"value": world * B * T * steps / el, "ms_per_step": 1e3 * el / steps, "transfer_only": probe}
        except Exception as e:
            res["copy"] = {"error": "%s: %s" % (type(e).__name__, e)}
    ok = {k: v for k, v in res.items() if "value" in v}
    fastest = max(ok, key=lambda k: ok[k]["value"]) if ok else None
    return {"value": ok[fastest]["value"] if fastest else None, "unit": "frames/s",
            "ms_per_step": ok[fastest]["ms_per_step"] if fastest else None, "transport": fastest, "steps": steps,
            "chunks": args.from_root_chunks, "global_batch": world * B, "bytes_per_peer_each_way": B * T * spec.FEATURE_DIM * 4,
            "transports": res,
            "note": "BatchShardedForward.forward_from_root: rank 0 holds [N*B,T,129,1] in HBM, every rank computes its batch slice, "
                    "the masks end up on rank 0; chunked so that transfer overlaps compute.  transports.rccl: slices scattered / "
                    "gathered by RCCL send/recv on two communicators (one peer per xGMI link); `value` there is measured with every CU "
                    "given to the forward (reserved_cus 0), the sweep leaves r CUs to RCCL's kernels.  transports.copy: the peers pull "
                    "/ push through IPC handles to rank 0's tensors with device-to-device copies (SDMA engines, no CU).  This entry's "
                    "value = the faster transport's.  Reported beside the headline `value`, not as it.  transfer_only: the same call "
                    "with no compute, one direction at a time (3 calls each): with the overlap working, ms_per_step ~ max(resident "
                    "ms_per_step, scatter_only_ms, gather_only_ms) + one chunk's transfer at each end."}


def secondary_config2(torch, build_model, spec, _lib, _weights, local_rank):
    """BASELINE configs[1]: R-CED V2 (16-layer) forward, batch 64, 129x512, bf16 (model_utils/model.py:32-61)."""
    B, T, steps, warmup = 64, 512, 200, 100   # a 0.5-ms kernel: 10 warm-up launches (5 ms) left the clocks where the CPU baseline's idle
                                              # seconds had put them (0.531 ms per launch against 0.496 in a run of its own, same box)
    w = _weights.synthetic_weights(2, seed=42)
    model = build_model("FullyCNNV2", False, weights=w, device=local_rank, dtype="bfloat16")
    args = None
    elapsed, times, _ = forward_line(args, torch, model, spec, _lib, 2, "bf16", B, T, steps, warmup, 1, 0, True)
    flops = spec.flops_per_frame(2) * B * T
    ms = 1e3 * elapsed / steps
    dom = max(times, key=lambda k: times[k][0])
    whole = {}
    for k in ("rced_fused", "rced_final_gemm"):
        for pipe, f in forward_flops_by_pipe(spec, 2, k, "bf16").items():
            whole[pipe] = whole.get(pipe, 0) + f * B * T
    pr = pipe_roofline(whole, ms * 1e-3)
    out = {"config": "R-CED V2 (16-layer) forward, batch 64, 129x512, bf16 activations/weights, fp32 accumulation "
                     "(BASELINE configs[1])",
           "metric": "spectrogram frames/sec (FullyCNNV2 fwd, 129-bin)", "value": B * T * steps / elapsed, "unit": "frames/s",
           "ms_per_step": ms, "steps": steps, "warmup": warmup, "dtype": "bf16",
           "tflops": flops / (ms * 1e-3) / 1e12,
           "roofline": dict(pr, frac_bf16_peak=flops / (ms * 1e-3) / 1e12 / BF16_PEAK_TFLOPS,
                            note="whole forward (nominal dense FLOPs) over wall time per step; one kernel, every layer on the plain bf16 "
                                 "MFMA (the input is cast to bf16, SURVEY 8 d2), so frac = frac_bf16_peak = TFLOP/s / 2500"),
           "kernels": {k: dict(pipe_roofline({p: f * B * T * steps for p, f in forward_flops_by_pipe(spec, 2, k, "bf16").items()}, v[0] * 1e-3),
                               avg_launch_ms=v[0] / v[1], launches=v[1]) for k, v in times.items() if v[1]},
           "kernels_ms_per_step": {k: v[0] / steps for k, v in times.items() if v[1]}, "dominant_kernel": dom}
    model.close()
    return out


def secondary_config5(torch, ge, FullyCNNTrainer, spec, _weights, local_rank):
    """BASELINE configs[4]: CR-CED V3 training step (fwd + bwd + Adam), batch 256 (model_utils/trainer.py:181-192)."""
    B, T, steps, warmup = 256, 512, 20, 3
    w = _weights.synthetic_weights(3, seed=42)
    tr = FullyCNNTrainer("FullyCNNV3", batch_size=B, lr=1e-3, warmup_steps=4000.0, weights=w, device=local_rank)
    x = torch.from_numpy(synthetic_magnitudes((B, T, 129, 1), 1234)).cuda()      # SURVEY 8(d2), C5 = C3 + target seed 1235
    y = torch.from_numpy(synthetic_magnitudes((B, T, 129, 1), 1235)).cuda()
    losses = []
    for _ in range(warmup):
        losses.append(tr.fit_step(x, y)[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(tr.fit_step(x, y)[0])      # rced_train_step synchronises (it returns the loss)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = 1e3 * elapsed / steps
    flops = 3 * spec.flops_per_frame(3) * B * T                 # forward + dgrad + wgrad, nominal
    free, total = torch.cuda.mem_get_info()
    pmc, pmc_src = pmc_record("pmc_train.json", ge.train_kernel_hash())
    pr = pipe_roofline({p: f * B * T for p, f in train_flops_by_pipe(spec).items()}, ms * 1e-3)
    out = {"config": "CR-CED V3 training step (fwd+bwd+Adam), batch 256, 129x512, fp32 (BASELINE configs[4])",
           "metric": "training step time", "value": ms, "unit": "ms/step", "higher_is_better": False,
           "ms_per_step": ms, "steps": steps, "warmup": warmup, "dtype": "f32",
           "frames_per_s": B * T * steps / elapsed, "tflops": flops / (ms * 1e-3) / 1e12,
           "roofline": dict(pr, note="3 x forward FLOPs (nominal) over wall time per step, priced on the pipes the step's "
                                      "convolutions run on (DESIGN 3.5); the step is layer-by-layer and also HBM-heavy "
                                      "(hbm_gb_per_step)"),
           "hbm_gb_per_step": (pmc or {}).get("hbm_gb_per_step"),
           "hbm_note": ("rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE summed over the step's kernels, %s" % pmc_src
                        if pmc else "no PMC capture for this kernel build"),
           "loss_first": losses[0], "loss_last": losses[-1], "device_mem_gb": (total - free) / 1e9}
    tr.close()
    return out


def secondary_rced_fp32(torch, build_model, spec, _lib, _weights, local_rank, variant):
    """R-CED V1 / V2 forward in fp32 at config 3's shape (batch 256, 129x512; model_utils/model.py:6-61): the other two
    nets of the path under the driver's clock, with one roofline entry per kernel."""
    B, T, steps, warmup = 256, 512, 30, 5
    name = NET_WORK[variant]
    w = _weights.synthetic_weights(variant, seed=42)
    model = build_model(name, False, weights=w, device=local_rank)
    elapsed, times, _ = forward_line(None, torch, model, spec, _lib, variant, "f32", B, T, steps, warmup, 1, 0, True)
    ms = 1e3 * elapsed / steps
    flops = spec.flops_per_frame(variant)
    final = 2 * spec.FEATURE_DIM * sum(l.kh * l.kw * l.cin * l.cout for l in spec.layers(variant)[-1:])
    per_kernel = {}
    for k, (tot, launches) in times.items():
        if not launches:
            continue
        kf = final if k == "rced_final_gemm" else flops - final
        pr = pipe_roofline({p: f * B * T * steps for p, f in forward_flops_by_pipe(spec, variant, k).items()}, tot * 1e-3)
        per_kernel[k] = dict(pr, avg_launch_ms=tot / launches, launches=launches, flop_per_frame=kf)
    out = {"config": "%s (%d-layer R-CED) forward, batch 256, 129x512, fp32 (config 3's shape; model.py:%s)"
                     % (name, len(spec.layers(variant)), "6-29" if variant == 1 else "32-61"),
           "metric": "spectrogram frames/sec (%s fwd, 129-bin)" % name, "value": B * T * steps / elapsed, "unit": "frames/s",
           "ms_per_step": ms, "steps": steps, "warmup": warmup, "dtype": "f32",
           "tflops": flops * B * T / (ms * 1e-3) / 1e12,
           "roofline": dict(pipe_roofline({p: sum(forward_flops_by_pipe(spec, variant, k).get(p, 0) for k in ("rced_fused", "rced_final_gemm")) * B * T
                                           for p in PIPE_CEILING_TFLOPS}, ms * 1e-3),
                            note="whole forward over wall time per step; per kernel (HIP events on the launch stream) below"),
           "kernels": per_kernel}
    model.close()
    return out


def secondary_config1_latency(torch, build_model, spec, _lib, _weights, local_rank, cpu_seconds=3.0):
    """BASELINE configs[0]: R-CED V1 forward on ONE 129x256 spectrogram the way infer.py calls it (infer.py:62-65):
    numpy in, numpy out through rced_forward_host -- a latency, not a throughput."""
    import numpy as np
    T, reps = 256, 200
    model = build_model("FullyCNN", False, weights=_weights.synthetic_weights(1, seed=42), device=local_rank)
    xh = synthetic_magnitudes((1, T, spec.FEATURE_DIM, 1), 1234)
    for _ in range(10):
        yh = model(xh)
    t0 = time.perf_counter()
    for _ in range(reps):
        yh = model(xh)
    host_ms = 1e3 * (time.perf_counter() - t0) / reps
    xd = torch.from_numpy(xh).cuda()
    for _ in range(10):
        model(xd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model(xd)
    torch.cuda.synchronize()
    dev_ms = 1e3 * (time.perf_counter() - t0) / reps
    # the same call on the throughput form's 3-frame tiles (86 workgroups instead of 256): what the latency form is worth
    model.set_option("latency_form", 0)
    for _ in range(10):
        model(xd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model(xd)
    torch.cuda.synchronize()
    dev3_ms = 1e3 * (time.perf_counter() - t0) / reps
    model.close()
    cpu = None
    if cpu_seconds > 0:     # BASELINE.md section 3 plans a CPU timing for C1 too: the torch-CPU restatement on the same spectrogram
        from oracle import torch_ref
        ref = torch_ref.TorchRef("FullyCNN", _weights.synthetic_weights(1, seed=42))
        xt = torch.from_numpy(xh)
        best, default_threads = None, torch.get_num_threads()
        for threads in sorted({1, 4, 8, 16, default_threads}):
            if threads > (os.cpu_count() or 1):
                continue
            torch.set_num_threads(threads)
            ref(xt)
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < cpu_seconds / 5:
                ref(xt)
                n += 1
            ms1 = 1e3 * (time.perf_counter() - t0) / n
            if best is None or ms1 < best[0]:
                best = (ms1, threads, n)
        torch.set_num_threads(default_threads)
        cpu = {"value": best[0], "unit": "ms", "cores": best[1], "kind": "port", "higher_is_better": False,
               "frames_per_s": T / (best[0] * 1e-3),
               "sample": "torch-CPU fp32 restatement (oracle/torch_ref.py) of R-CED V1 on the same [1,256,129,1] spectrogram, "
                         "%d calls on %d threads (the fastest of 1/4/8/16/default threads; host has %d logical cpus)"
                         % (best[2], best[1], os.cpu_count() or 1)}
    return {"config": "R-CED V1 (10-layer) forward on one 129x256 spectrogram, numpy in -> numpy out (BASELINE configs[0], "
                      "infer.py:62-65)", "cpu_baseline": cpu,
            "metric": "latency per utterance", "value": host_ms, "unit": "ms", "higher_is_better": False,
            "ms_per_step": host_ms, "steps": reps, "dtype": "f32", "frames_per_s": T / (host_ms * 1e-3),
            "device_resident_ms": dev_ms, "device_resident_ms_3frame_tiles": dev3_ms, "finite": bool(np.isfinite(yh).all()),
            "note": "host path = H2D + fused kernel + final GEMM + D2H + synchronise per call; device_resident_ms = the two "
                    "launches alone, back to back (one tile's trip through all layers: launch-bound, no roofline claim).  The call "
                    "runs the kernel's latency form (one-frame tiles: 256 workgroups for the 256 frames; option latency_form); "
                    "device_resident_ms_3frame_tiles = the same on the throughput form's tiles (86 workgroups)"}


def secondary_pipeline(torch, build_model, spec, _lib, _weights, local_rank, cpu_seconds):
    """PCM -> STFT -> CR-CED -> ISTFT -> PCM at config-3 scale (infer.py:54-71 as a batch; data_utils/audio_feature.py:22-44,
    model_utils/utils.py:171-183): rced_stft / rced_forward / rced_istft device-resident, the three kernel times, and the
    CPU leg for the two audio stages (the reference's own numpy algorithm, oracle/audio_np) beside them."""
    import numpy as np
    from fullycnnspeechenhancement_amd import audio
    N, T, reps = 256, 512, 10
    L = (T - 1) * 128 + 256
    rng = np.random.default_rng(4321)
    pcm_h = (rng.standard_normal((N, L), dtype=np.float32) * np.float32(0.1))
    pcm = torch.from_numpy(pcm_h).cuda()
    model = build_model("FullyCNNV3", False, weights=_weights.synthetic_weights(3, seed=42), device=local_rank)
    model.reserve(N, T)

    def timed(fn):
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # the kernels run on torch's current stream
        a.record()
        for _ in range(reps):
            out = fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps, out

    t_stft, (mag, ph) = timed(lambda: audio.stft_batch(pcm))
    assert int(mag.shape[1]) == T
    t_cnn, pred = timed(lambda: model(mag))
    t_istft, wav = timed(lambda: audio.istft_batch(pred, ph))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m2, p2 = audio.stft_batch(pcm)
        wav = audio.istft_batch(model(m2), p2)
    torch.cuda.synchronize()
    whole = 1e3 * (time.perf_counter() - t0) / reps
    frames = N * T
    flop_dft = 2 * 256 * 258
    audio_pipe = "mfma_bf16x6"     # audio.stft_batch / istft_batch launch the three-part bf16 kernels (kernels="f32": the fp32-MFMA comparators)
    out = {"config": "PCM -> STFT -> CR-CED V3 -> ISTFT -> PCM, 256 utterances x 65,664 samples (512 frames), device-resident "
                     "(SURVEY 8(f) N1 + a5 + N2; infer.py:54-71)",
           "metric": "spectrogram frames/sec through the whole pipeline", "value": frames / (whole * 1e-3), "unit": "frames/s",
           "ms_per_step": whole, "steps": reps, "dtype": "f32",
           "kernels_ms": {"rced_stft": t_stft, "rced_forward": t_cnn, "rced_istft": t_istft},
           "stft": dict(pipe_roofline({audio_pipe: frames * flop_dft}, t_stft * 1e-3), tflops=frames * flop_dft / t_stft / 1e9,
                        algorithmic_gbps=(N * L * 4 + frames * 129 * 12) / t_stft / 1e6,
                        frac_hbm_8tbs=(N * L * 4 + frames * 129 * 12) / t_stft / 1e6 / 8000.0),
           "istft": dict(pipe_roofline({audio_pipe: frames * flop_dft}, t_istft * 1e-3), tflops=frames * flop_dft / t_istft / 1e9,
                         algorithmic_gbps=(frames * 129 * 12 + N * (T + 1) * 128 * 4) / t_istft / 1e6,
                         frac_hbm_8tbs=(frames * 129 * 12 + N * (T + 1) * 128 * 4) / t_istft / 1e6 / 8000.0,
                         note="nominal FLOPs of the full 256 x 258 inverse transform; de_frame keeps half of its rows, the kernel "
                              "computes only those (and de_emphasis in the same pass)"),
           "finite": bool(torch.isfinite(wav).all()),
           "note": "dense-DFT GEMMs (K = 256) in the three-part bf16 form, one M-tile per wave with its fragments in registers "
                   "(kernels_audio_x6.h; rced_stft_ex / rced_istft_ex with RCED_AUDIO_F32 run the fp32-MFMA comparators): 0.17 ms kernels, reported against the "
                   "matrix pipe and against HBM"}
    if cpu_seconds > 0:
        from oracle import audio_np
        done, t_st, t_is = 0, 0.0, 0.0
        t_begin = time.perf_counter()
        while time.perf_counter() - t_begin < cpu_seconds and done < N:
            t0 = time.perf_counter()
            mg, phs = audio_np.stft(pcm_h[done])
            t1 = time.perf_counter()
            audio_np.rebuild(mg, phs, nfft=512)
            t2 = time.perf_counter()
            t_st += t1 - t0
            t_is += t2 - t1
            done += 1
        out["cpu_baseline"] = {"kind": "port", "cores": 1, "unit": "frames/s",
                               "stft": done * T / t_st, "istft": done * T / t_is, "value": done * T / (t_st + t_is),
                               "sample": "oracle/audio_np.py (numpy restatement of audio_feature.py:22-44 and utils.py:171-183, "
                                         "pinned to the reference's own outputs; its de-emphasis is the reference's per-sample "
                                         "Python loop) on %d utterances of %d samples, one thread: %.2f s STFT, %.2f s rebuild"
                                         % (done, L, t_st, t_is)}
    model.close()
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args)          # does not return

    import __graft_entry__ as ge
    ge.build_hip()                  # before any torch.cuda / HIP call of this process (content-hash gated, locked)

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:   # before anything can fail: the launcher test reads this line (each rank prints its own)
        print("[bench] rank %d of WORLD_SIZE=%d started" % (rank, world), file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    if world != max(args.gpus, 1):
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run --nproc-per-node %d "
                         "(or bare `python bench.py --gpus %d`, which starts the ranks itself)"
                         % (args.gpus, world, args.gpus, args.gpus))
    # RCED_BENCH_REHEARSE=1 (tests only, labelled in the line): the N > 1 control flow on a box with fewer GPUs than ranks --
    # ranks share the GPUs round-robin and the control plane is gloo (RCCL refuses two ranks on one device).  What it
    # exercises is this file's N > 1 branch (launch, barriers, max over ranks, the line); it measures nothing.
    rehearse = world > 1 and os.environ.get("RCED_BENCH_REHEARSE", "0") == "1"
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if rehearse else local_rank
    ctl = "cpu" if rehearse else "cuda"
    torch.cuda.set_device(dev_index)
    rccl_world = 1
    if world > 1:
        import datetime
        if rehearse:
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=4))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=4))
        ones = torch.ones(1, device=ctl)
        dist.all_reduce(ones)                      # the rank count RCCL itself sees
        rccl_world = int(ones.item())

    from fullycnnspeechenhancement_amd import FullyCNNTrainer, _lib, build_model, spec, weights as _weights
    # (oracle/ is imported only inside cpu_baseline(): it is the checker / CPU baseline, never the measured path)

    variant = args.variant
    weights = _weights.synthetic_weights(variant, seed=42)                # random-init, SURVEY 8(d2)
    model = build_model(NET_WORK[variant], False, weights=weights, device=dev_index,
                        dtype="bfloat16" if args.dtype == "bf16" else "float32")
    model.set_path(args.path)
    B, T = args.batch, args.frames
    elapsed, times, (x, y) = forward_line(args, torch, model, spec, _lib, variant, args.dtype, B, T, args.steps,
                                          args.warmup, world, rank, not args.no_profile, ctl)

    frames_total = world * B * T * args.steps
    flops_frame = spec.flops_per_frame(variant)
    out = {
        "metric": "spectrogram frames/sec (CR-CED-16 fwd, 129-bin)" if variant == 3 else
                  "spectrogram frames/sec (%s fwd, 129-bin)" % NET_WORK[variant],
        "value": frames_total / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "CR-CED V3 (16-layer, skip connections) forward, batch %d per GPU, 129x%d, fp32 "
                               "(BASELINE configs[2])" % (B, T) if variant == 3 else
                               "%s forward, batch %d per GPU, 129x%d, %s" % (NET_WORK[variant], B, T, "fp32" if args.dtype == "f32" else "bf16 activations/weights, fp32 accumulation"),
                   "variant": NET_WORK[variant], "batch_per_gpu": B, "frames": T, "bins": spec.FEATURE_DIM,
                   "global_batch": world * B, "path": {0: "auto", 1: "layerwise", 2: "fused"}[model.get_option("path")],
                   "fused_available": bool(model.get_option("has_fused")), "parallelism": "batch-shard x%d" % world,
                   "rccl_world_size": rccl_world, "weights": "random-init (glorot, seed 42)",
                   "input": "|N(0,1)| float32, numpy default_rng(1234 + rank), resident in HBM before the timed region",
                   "library": _lib.version()},
        "from_root": None,
    }
    if rehearse:
        out["config"]["rehearsal"] = ("RCED_BENCH_REHEARSE=1: %d ranks on %d GPU(s), gloo control plane -- a test of the N > 1 "
                                      "control flow, NOT a measurement" % (world, torch.cuda.device_count()))
    if rank == 0:
        roof = None
        if times is not None:
            dom = max(times, key=lambda k: times[k][0])
            ms, launches = times[dom]
            if launches:
                # FLOPs the dominant kernel kind performs per forward (nominal dense count)
                final = 2 * spec.FEATURE_DIM * sum(l.kh * l.kw * l.cin * l.cout for l in spec.layers(variant)[-1:])
                fused_all = bool(model.get_option("fused_final"))     # the output layer runs inside the fused kernel (CR-CED; R-CED in bf16)
                if dom == "conv_layer_generic" or (dom == "rced_fused" and fused_all):
                    kflops = flops_frame
                else:
                    kflops = final if dom == "rced_final_gemm" else flops_frame - final
                achieved = kflops * B * T * args.steps / (ms * 1e-3) / 1e12
                hand_ch = spec.layers(variant)[-1].cin     # channels of the tensor handed to the final 1x129 layer
                traffic, traffic_src = pmc_traffic(ge, variant, B, T, dom)
                alg = 1032 * B * T if (dom != "rced_fused" or fused_all) else (516 + 516 * hand_ch) * B * T
                v3x6 = int(model.get_option("v3_l2x6")) if variant == 3 else 0
                pr = pipe_roofline({p_: f * B * T * args.steps
                                    for p_, f in forward_flops_by_pipe(spec, variant, dom, args.dtype, v3x6).items()}, ms * 1e-3)
                roof = dict(pr, kernel=dom, traffic=traffic,
                            traffic_note="HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE: %s; this kernel's "
                                         "algorithmic bytes per launch = %d; whole forward = %d" % (traffic_src, alg, 1032 * B * T),
                            avg_launch_ms=ms / launches, floor_ms_per_launch=pr["frac"] * ms / launches, launches=launches,
                            flop_per_frame=kflops, frames_per_forward=B * T,
                            other_kernels_ms_per_step={k: v[0] / args.steps for k, v in times.items() if k != dom and v[1]},
                            note="compute-bound path (7950 FLOP/B): bound = matrix issue, not HBM.  CR-CED: every layer is computed at "
                                 "fp32 quality as six bf16 MFMAs per product over three-part operands (DESIGN 3.1; pipe_mix says "
                                 "which share when another form of the kernel is selected).  `peak` = the rate this mix of pipes reaches with every "
                                 "pipe at the guide's spec peak (157.3 fp32 MFMA; 2500 / 6 for the six-product form), `frac` = "
                                 "achieved / peak = (sum FLOP_i / peak_i) / avg launch time; frac_fp32_peak = achieved / 157.3 "
                                 "(SURVEY 8(d3)'s figure, above 1 for a kernel on the faster pipe); frac_measured_ceiling prices "
                                 "the six-product form at the 386 TFLOP/s a register-only loop reaches; algorithmic HBM bytes "
                                 "are 1032 B/frame")
                assert abs(roof["achieved"] - achieved) < 1e-6 * achieved
        out["roofline"] = roof
    # ---- the reference's single-host-process convention over RCCL: scatter from rank 0, compute, gather -------
    # The headline figures above are complete at this point.  from_root is the first code of a run that sends utterances
    # between GPUs; if it stalls (a link, a communicator that never forms), the collective would sit until the process
    # group's own watchdog ABORTS every rank -- and the line with it.  So a timer stands beside it: on expiry rank 0 prints
    # the line as it is (from_root.error = the timeout) and every rank EXITS (never restarts or re-execs) with
    # --from-root-fail-status, 3 by default: a stalled scatter must not read as success to a caller that gates on the
    # status.  The line is on stdout before the exit, and `launch_ranks` relays it whatever the status.
    if world > 1 and args.from_root_steps > 0:
        import threading

        def give_up():
            if rank == 0:
                out["from_root"] = {"error": "timeout: forward_from_root did not finish within %d s; the headline figures "
                                             "were complete before it started" % args.from_root_timeout}
                out.setdefault("host_buffers", None)
                out.setdefault("cpu_baseline", None)
                print(json.dumps(out), flush=True)
            sys.stderr.write("[bench] rank %d: from_root timed out after %d s, leaving with status %d\n"
                             % (rank, args.from_root_timeout, args.from_root_fail_status))
            sys.stderr.flush()
            os._exit(args.from_root_fail_status)

        timer = threading.Timer(args.from_root_timeout, give_up)
        timer.daemon = True
        timer.start()
        try:
            if rehearse:
                if os.environ.get("RCED_BENCH_REHEARSE_HANG", "0") == "1":     # tests: a from_root that never returns
                    while True:
                        time.sleep(1.0)
                # gloo does not move device tensors between ranks; the copy transport needs no communicator: the ranks share this box's
                # GPU(s) and the IPC-handle path runs for real
                out["from_root"] = from_root_line(args, torch, dist, model, spec, world, rank, dev_index, B, T, ctl="cpu", transports=("copy",))
            else:
                out["from_root"] = from_root_line(args, torch, dist, model, spec, world, rank, dev_index, B, T)
        except Exception as e:      # the headline line must survive a failure of the secondary figure
            out["from_root"] = {"error": "%s: %s" % (type(e).__name__, e)}
        timer.cancel()
    host_line = None
    if rank == 0 and world == 1 and not args.no_secondary:
        try:
            host_line = host_buffers_line(model, x)
        except Exception as e:
            host_line = {"error": "%s: %s" % (type(e).__name__, e)}
    out["host_buffers"] = host_line
    del x, y
    model.close()
    if rank == 0:
        # the CPU restatement and the secondary configs run on rank 0 of the 1-GPU run only
        out["cpu_baseline"] = (cpu_baseline(variant, weights, T, args.cpu_seconds)
                               if args.cpu_seconds > 0 and world == 1 else None)
        if world == 1 and not args.no_secondary:
            sec = []
            for fn, a in ((secondary_config2, (torch, build_model, spec, _lib, _weights, local_rank)),
                          (secondary_config5, (torch, ge, FullyCNNTrainer, spec, _weights, local_rank)),
                          (secondary_rced_fp32, (torch, build_model, spec, _lib, _weights, local_rank, 1)),
                          (secondary_rced_fp32, (torch, build_model, spec, _lib, _weights, local_rank, 2)),
                          (secondary_config1_latency, (torch, build_model, spec, _lib, _weights, local_rank,
                                                       min(args.cpu_seconds, 3.0))),
                          (secondary_pipeline, (torch, build_model, spec, _lib, _weights, local_rank,
                                                min(args.cpu_seconds, 6.0)))):
                try:
                    sec.append(fn(*a))
                except Exception as e:      # a failing secondary must not take the headline line down with it
                    sec.append({"config": fn.__doc__.split(":")[0].strip(), "error": "%s: %s" % (type(e).__name__, e)})
                torch.cuda.empty_cache()
            out["secondary"] = sec
        print(json.dumps(out), flush=True)
    if world > 1:
        # the line is out; a peer that is gone must not turn the run into a watchdog abort while everybody says goodbye
        import threading
        bye = threading.Timer(60.0, lambda: os._exit(args.from_root_fail_status))   # a peer never reached the closing barrier
        bye.daemon = True
        bye.start()
        dist.barrier()
        dist.destroy_process_group()
        bye.cancel()


if __name__ == "__main__":
    main()
