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
(BASELINE configs 2 and 5 under the same clock).
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
NET_WORK = {1: "FullyCNN", 2: "FullyCNNV2", 3: "FullyCNNV3"}


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
    bounded sample of the same workload: batches of 8 utterances x T frames until ~budget_s.  oneDNN does
    not always scale to every hardware thread, so a short probe picks the fastest thread count first."""
    import torch
    from oracle import rced_np, torch_ref
    ref = torch_ref.TorchRef(NET_WORK[variant], weights)
    x = torch.from_numpy(rced_np.make_input(8, frames_t, seed=1234))
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    cands = sorted({default_threads} | {c for c in (8, 16, 32, 64, 128, 256) if c <= ncpu})
    probe, best = {}, default_threads
    for c in cands:
        torch.set_num_threads(c)
        ref(x[:1])
        t0 = time.perf_counter()
        ref(x)
        probe[c] = x.shape[0] * x.shape[1] / (time.perf_counter() - t0)
        if probe[c] > probe.get(best, 0):
            best = c
    torch.set_num_threads(best)
    ref(x[:1])  # warm-up (oneDNN primitive creation)
    done, t0 = 0, time.perf_counter()
    while True:
        ref(x)
        done += x.shape[0] * x.shape[1]
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    torch.set_num_threads(default_threads)
    return {"value": done / el, "unit": "frames/s", "cores": int(best), "kind": "port",
            "sample": "torch-CPU fp32 restatement (oracle/torch_ref.py), %d frames = %d batches of [8,%d,129,1] in %.1f s "
                      "on %d threads (fastest of %s); host has %d logical cpus"
                      % (done, done // (8 * frames_t), frames_t, el, best, sorted(probe), ncpu)}


def pmc_record(name, kernel_hash):
    """A committed rocprofv3 PMC capture (profiles/<name>), or None when it was taken on other kernel code than the
    one being run: every capture records the hash of the sources that define the kernels and their packed-weight /
    LDS layouts (__graft_entry__.forward_kernel_hash / train_kernel_hash)."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            d = json.load(fh)
        return d if d.get("kernel_hash") == kernel_hash else None
    except Exception:
        return None


def pmc_traffic(ge, variant, batch, frames, kernel):
    """HBM bytes per launch of the dominant kernel (FETCH_SIZE + WRITE_SIZE, separate passes, corrected as
    MI355X_MICROARCH.md prescribes; collected by tools/profile.sh on this same command)."""
    d = pmc_record("r02_pmc_traffic.json", ge.forward_kernel_hash())
    if not d:
        return None, "no PMC capture for this kernel build (profiles/r02_pmc_traffic.json records another kernel_hash)"
    wl = d["workload"]
    key = {"rced_fused": "fused_v3_kernel", "rced_final_gemm": "final_gemm_kernel"}.get(kernel)
    if key in d and (wl["variant"], wl["batch"], wl["frames"]) == (variant, batch, frames):
        return 1024 * (d[key]["fetch_kib"] + d[key]["write_kib"]), "profiles/r02_pmc_traffic.json (kernel_hash %s)" % d["kernel_hash"]
    return None, "PMC capture is for another workload"


def synthetic_magnitudes(shape, seed):
    """x = |N(0,1)| float32 from numpy.random.default_rng(seed) (magnitude spectrograms are non-negative)."""
    import numpy as np
    x = np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)
    return np.abs(x, out=x)


def forward_line(args, torch, model, spec, _lib, variant, dtype, B, T, steps, warmup, world, rank, profile):
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
        tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
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
        yh = model(xh)             # a fresh output array per call, as sess.run returns one
    el = time.perf_counter() - t0
    yo = np.empty_like(xh)
    model(xh, out=yo)
    t0 = time.perf_counter()
    for _ in range(steps):
        model(xh, out=yo)          # the caller's output array reused: no page faults of a fresh 67 MB buffer
    el_out = time.perf_counter() - t0
    n, t = xh.shape[0], xh.shape[1]
    return {"value": n * t * steps / el, "unit": "frames/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
            "ms_per_step_reused_output": 1e3 * el_out / steps, "same_result": bool(np.array_equal(yh, yo)),
            "bytes_each_way": int(xh.nbytes), "finite": bool(abs(float(yh.sum())) < float("inf")),
            "note": "numpy [N,T,129,1] in -> numpy out through rced_forward_host: H2D + kernel + D2H per call, "
                    "PCIe-inclusive (SURVEY 8(d2) 'with-H2D/D2H figure'); never `value`.  ms_per_step: a fresh output ndarray per call, as "
                    "sess.run returns one (the OS zero-fills its 67 MB: ~5 ms of page faults); ms_per_step_reused_output: "
                    "model(x, out=buf)"}


def from_root_line(args, torch, dist, model, spec, world, rank, local_rank, B, T):
    """The reference's single-host-process convention over RCCL: rank 0 holds the whole batch, scatters batch slices,
    every rank computes, the masks gather back (fullycnnspeechenhancement_amd/dist.py)."""
    from fullycnnspeechenhancement_amd.dist import BatchShardedForward
    eng = BatchShardedForward(model, device="cuda:%d" % local_rank, forward_into=lambda a, out: model(a, out=out))
    xr = None
    if rank == 0:
        xr = torch.from_numpy(synthetic_magnitudes((world * B, T, spec.FEATURE_DIM, 1), 1234)).cuda()
    for _ in range(2):
        eng.forward_from_root(xr, root=0, chunks=args.from_root_chunks)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.from_root_steps):
        eng.forward_from_root(xr, root=0, chunks=args.from_root_chunks)
    torch.cuda.synchronize()
    dist.barrier()
    el = time.perf_counter() - t0
    tmax = torch.tensor([el], device="cuda", dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    el = float(tmax.item())
    return {"value": world * B * T * args.from_root_steps / el, "unit": "frames/s",
            "ms_per_step": 1e3 * el / args.from_root_steps, "steps": args.from_root_steps,
            "chunks": args.from_root_chunks, "global_batch": world * B,
            "bytes_per_peer_each_way": B * T * spec.FEATURE_DIM * 4,
            "note": "BatchShardedForward.forward_from_root: rank 0 holds [N*B,T,129,1] in HBM, scatters batch "
                    "slices over RCCL send/recv (one peer per xGMI link), every rank computes, masks gather back "
                    "to rank 0; chunked so that transfer overlaps compute.  Reported beside `value`, not as it."}


def secondary_config2(torch, build_model, spec, _lib, _weights, local_rank):
    """BASELINE configs[1]: R-CED V2 (16-layer) forward, batch 64, 129x512, bf16 (model_utils/model.py:32-61)."""
    B, T, steps, warmup = 64, 512, 50, 10
    w = _weights.synthetic_weights(2, seed=42)
    model = build_model("FullyCNNV2", False, weights=w, device=local_rank, dtype="bfloat16")
    args = None
    elapsed, times, _ = forward_line(args, torch, model, spec, _lib, 2, "bf16", B, T, steps, warmup, 1, 0, True)
    flops = spec.flops_per_frame(2) * B * T
    ms = 1e3 * elapsed / steps
    dom = max(times, key=lambda k: times[k][0])
    out = {"config": "R-CED V2 (16-layer) forward, batch 64, 129x512, bf16 activations/weights, fp32 accumulation "
                     "(BASELINE configs[1])",
           "metric": "spectrogram frames/sec (FullyCNNV2 fwd, 129-bin)", "value": B * T * steps / elapsed, "unit": "frames/s",
           "ms_per_step": ms, "steps": steps, "warmup": warmup, "dtype": "bf16",
           "tflops": flops / (ms * 1e-3) / 1e12,
           "roofline": {"bound": "mfma", "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "achieved": flops / (ms * 1e-3) / 1e12, "frac": flops / (ms * 1e-3) / 1e12 / BF16_PEAK_TFLOPS,
                        "note": "whole forward (nominal dense FLOPs) over wall time per step vs the dense bf16 MFMA peak"},
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
    pmc = pmc_record("r02_pmc_train.json", ge.train_kernel_hash())
    out = {"config": "CR-CED V3 training step (fwd+bwd+Adam), batch 256, 129x512, fp32 (BASELINE configs[4])",
           "metric": "training step time", "value": ms, "unit": "ms/step", "higher_is_better": False,
           "ms_per_step": ms, "steps": steps, "warmup": warmup, "dtype": "f32",
           "frames_per_s": B * T * steps / elapsed, "tflops": flops / (ms * 1e-3) / 1e12,
           "roofline": {"bound": "mfma", "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "achieved": flops / (ms * 1e-3) / 1e12, "frac": flops / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                        "note": "3 x forward FLOPs (nominal) over wall time per step; the step is layer-by-layer and also "
                                "HBM-heavy (hbm_gb_per_step)"},
           "hbm_gb_per_step": (pmc or {}).get("hbm_gb_per_step"),
           "hbm_note": ("rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE summed over the step's kernels, profiles/r02_pmc_train.json"
                        if pmc else "no PMC capture for this kernel build"),
           "loss_first": losses[0], "loss_last": losses[-1], "device_mem_gb": (total - free) / 1e9}
    tr.close()
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
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    if world != max(args.gpus, 1):
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run --nproc-per-node %d "
                         "(or bare `python bench.py --gpus %d`, which starts the ranks itself)"
                         % (args.gpus, world, args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    rccl_world = 1
    if world > 1:
        import datetime
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=4))
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)                      # the rank count RCCL itself sees
        rccl_world = int(ones.item())

    from fullycnnspeechenhancement_amd import FullyCNNTrainer, _lib, build_model, spec, weights as _weights
    # (oracle/ is imported only inside cpu_baseline(): it is the checker / CPU baseline, never the measured path)

    variant = args.variant
    weights = _weights.synthetic_weights(variant, seed=42)                # random-init, SURVEY 8(d2)
    model = build_model(NET_WORK[variant], False, weights=weights, device=local_rank,
                        dtype="bfloat16" if args.dtype == "bf16" else "float32")
    model.set_path(args.path)
    B, T = args.batch, args.frames
    elapsed, times, (x, y) = forward_line(args, torch, model, spec, _lib, variant, args.dtype, B, T, args.steps,
                                          args.warmup, world, rank, not args.no_profile)

    # ---- the reference's single-host-process convention over RCCL: scatter from rank 0, compute, gather -------
    from_root = None
    if world > 1 and args.from_root_steps > 0:
        try:
            from_root = from_root_line(args, torch, dist, model, spec, world, rank, local_rank, B, T)
        except Exception as e:      # the headline line must survive a failure of the secondary figure
            from_root = {"error": "%s: %s" % (type(e).__name__, e)}

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
        "from_root": from_root,
    }
    if rank == 0:
        roof = None
        if times is not None:
            dom = max(times, key=lambda k: times[k][0])
            ms, launches = times[dom]
            if launches:
                # FLOPs the dominant kernel kind performs per forward (nominal dense count)
                final = 2 * spec.FEATURE_DIM * sum(l.kh * l.kw * l.cin * l.cout for l in spec.layers(variant)[-1:])
                fused_all = bool(model.get_option("fused_final")) if variant == 3 else False
                if dom == "conv_layer_generic" or (dom == "rced_fused" and fused_all):
                    kflops = flops_frame
                else:
                    kflops = final if dom == "rced_final_gemm" else flops_frame - final
                achieved = kflops * B * T * args.steps / (ms * 1e-3) / 1e12
                hand_ch = spec.layers(variant)[-1].cin     # channels of the tensor handed to the final 1x129 layer
                traffic, traffic_src = pmc_traffic(ge, variant, B, T, dom)
                peak = FP32_PEAK_TFLOPS if args.dtype == "f32" else BF16_PEAK_TFLOPS
                alg = 1032 * B * T if (dom != "rced_fused" or fused_all) else (516 + 516 * hand_ch) * B * T
                roof = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": peak,
                        "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                        "traffic_note": "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE: %s; this kernel's "
                                        "algorithmic bytes per launch = %d; whole forward = %d" % (traffic_src, alg, 1032 * B * T),
                        "avg_launch_ms": ms / launches, "launches": launches,
                        "flop_per_frame": kflops, "frames_per_forward": B * T,
                        "other_kernels_ms_per_step": {k: v[0] / args.steps for k, v in times.items() if k != dom and v[1]},
                        "note": "compute-bound path (7950 FLOP/B): peak = dense fp32 157.3 TFLOP/s, not HBM; "
                                "algorithmic HBM bytes are 1032 B/frame"}
        out["roofline"] = roof
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
                          (secondary_config5, (torch, ge, FullyCNNTrainer, spec, _weights, local_rank))):
                try:
                    sec.append(fn(*a))
                except Exception as e:      # a failing secondary must not take the headline line down with it
                    sec.append({"config": fn.__doc__.split(":")[0].strip(), "error": "%s: %s" % (type(e).__name__, e)})
                torch.cuda.empty_cache()
            out["secondary"] = sec
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
