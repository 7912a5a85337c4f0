#!/usr/bin/env python3
"""DGViT hot-path benchmark (contract: see the task statement / DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

Workload (BASELINE.json metric, config C3): per GPU a batch of 512 synthetic 84x84 depth frames + polar goals
through DGViT-small (GoTPolicy: 84x84 @ 12x12 patches, 6 layers, 8 heads, d=256, MLP 2048), train mode,
forward + backward of an actor loss on (mean, log_std) against random targets, gradient all-reduce over
RCCL when N > 1, then an Adam step.  One "step" = one such pass over one batch; value = frames/s over all
ranks.  Inputs are resident in HBM before the timed region.  fp32 throughout (v_mfma_f32_32x32x2_f32).

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# --grad-overlap: the HIP runtime multiplexes streams onto 4 hardware queues by default; with RCCL initialised, the stream the collectives
# run on then shares a queue with the compute stream and an all-reduce queued behind a gradient-ready event still executes after the whole
# backward (measured: tools/overlap_queue_probe.py, profiles/r03_d_overlap_queue_probe.txt).  Must be set before the first HIP call.
if "--grad-overlap" in sys.argv:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 256 CUs @ 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA dense (the 2:1-sparsity headline figure is never used)
IMAGE, PATCH, DIM, DEPTH, HEADS = (84, 84), (12, 12), 256, 6, 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=256, help="frames of the 512-frame batch the CPU baseline runs (fwd+bwd, best of 3)")
    ap.add_argument("--no-sac-step", action="store_true", help="skip the secondary full SAC-style step measurement")
    ap.add_argument("--profile-stride", type=int, default=10, help="time every n-th launch of each kernel kind with HIP events (1 = every launch)")
    ap.add_argument("--no-overlap-ab", action="store_true", help="skip the forward-only pass and the helper-stream A/B after the timed region (use when profiling: they launch the same kernels)")
    ap.add_argument("--no-c5", action="store_true", help="skip the secondary config-5 (224x224 ViT-Base, bf16) forward measurement")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the secondary launch-bound measurements (single-frame sample(), shipped learn() step)")
    ap.add_argument("--wgrad-overlap", action="store_true", help="A/B: weight-gradient GEMMs on the helper stream (+5%% frames/s, blurs per-kernel timing)")
    ap.add_argument("--dense-last-block", action="store_true", help="A/B: compute the last block for every token")
    ap.add_argument("--force-collective", action="store_true",
                    help="rehearsal of the N > 1 path on one GPU: initialise the RCCL process group even for one rank and run the "
                         "gradient all-reduce (a one-rank all-reduce is the identity) inside every step")
    ap.add_argument("--grad-overlap", action="store_true",
                    help="start every transformer block's gradient all-reduce from inside the backward, behind its gradient-ready event "
                         "(GradSync(overlap=True)); default: one exchange after the backward -- the overlapped form is covered by tests and a "
                         "one-card timeline only, it has never run across real GPUs (DESIGN section 6)")
    return ap.parse_args()


def cpu_baseline(batch):
    """The CPU oracle (same math as the reference's PyTorch-CPU path, pinned to it by tests/golden) on a bounded sample: `batch` frames
    fwd+bwd, best of 3.  BASELINE.md section 4 asks for the box's host cores: on a many-core host torch's intra-op pool is not fastest
    with one thread per core (256 threads on the 256-core box of round 4: 2.5 frames/s against 230 with 16), so a short probe picks the
    fastest of {16, 32, 64, every core this process may run on} threads and the sample runs with that; both figures are reported."""
    from oracle import dgvit_oracle as O
    cfg = O.GoTConfig(image=IMAGE, patch=PATCH, dim=DIM, depth=DEPTH, heads=HEADS)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, cores)
    params = O.make_params(O.policy_param_spec(cfg), 3407)
    for v in params.values():
        v.requires_grad_(True)

    def run(nframes, passes):
        img, pstate, _, _ = O.make_inputs(cfg, nframes, 3407)
        tm, tl = torch.randn(nframes, 2), torch.randn(nframes, 2)
        mask = (torch.rand(nframes, cfg.tokens, cfg.dim) < 0.9).float()
        best = float("inf")
        for i in range(passes + 1):
            t0 = time.perf_counter()
            mean, log_std = O.policy_forward(params, img, pstate, cfg, drop_mask=mask)
            loss = ((mean - tm) ** 2).mean() + ((log_std - tl) ** 2).mean()
            torch.autograd.grad(loss, [v for v in params.values()], allow_unused=True)
            dt = time.perf_counter() - t0
            if i > 0:
                best = min(best, dt)
        return nframes / best

    probe, skipped = {}, []
    for n in sorted({min(cores, 16), min(cores, 32), min(cores, 64), cores}):
        if probe and probe[max(probe)] < 0.5 * max(probe.values()):     # already past the knee: more threads only add contention
            skipped.append(n)                                             # (256 threads ran 0.2 frames/s here: 80 s for the 16-frame probe)
            continue
        torch.set_num_threads(n)
        probe[n] = run(16, 1)
    threads = max(probe, key=probe.get)
    torch.set_num_threads(threads)
    fps = run(batch, 3)
    return {"value": round(fps, 2), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle/dgvit_oracle.py policy fwd+bwd on {batch} of the 512 frames, train-mode mask, best of 3 after one warm-up pass, "
                      f"{threads} torch threads = the fastest of a 16-frame probe over {{threads: frames/s}} "
                      f"{ {k: round(v, 1) for k, v in probe.items()} }"
                      + (f" (not probed: {skipped} threads, the rate had already halved)" if skipped else "")
                      + f" on a host with {cores} usable cores ({os.cpu_count()} in the machine)"}


def sac_step(dgvit_amd, synthetic, B, dev, steps=5):
    """Secondary number (SURVEY 8(d) C3 "secondary"): one SAC-style update with transformer actor AND transformer
    critic, losses as DRL.py:390-432 (alpha fixed 0.2, gamma 0.99): 2 no-grad target passes, critic fwd+bwd,
    actor fwd + critic fwd + bwd through both, two Adam steps, Polyak update.  5 encoder forwards and 3 encoder
    backwards per frame."""
    from dgvit_amd.optim import FlatAdam, flatten_parameters, soft_update
    import copy
    torch.manual_seed(1)
    kw = dict(image_size=IMAGE, patch_size=PATCH)
    pol = dgvit_amd.GoTPolicy(2, 2, DEPTH, HEADS, DIM, **kw).to(dev)
    crt = dgvit_amd.GoTQNetwork(2, 2, DEPTH, HEADS, DIM, **kw).to(dev)
    tgt = copy.deepcopy(crt)
    flatten_parameters(crt), flatten_parameters(tgt)
    opt_p, opt_c = FlatAdam([pol], lr=1e-4), FlatAdam([crt], lr=1e-4)
    img, pstate, act, _ = (t.to(dev) for t in synthetic.make_inputs(IMAGE, B, 11))
    nimg, npst, _, _ = (t.to(dev) for t in synthetic.make_inputs(IMAGE, B, 12))
    rew = torch.randn(B, 1, device=dev)
    alpha, gamma, tau = 0.2, 0.99, 0.005

    def step():
        with torch.no_grad():
            na, nlogp, _ = pol.sample([nimg, npst])
            q1n, q2n = tgt([nimg, npst, na])
            y = rew + gamma * (torch.min(q1n, q2n) - alpha * nlogp)
        q1, q2 = crt([img, pstate, act])
        qf = torch.nn.functional.mse_loss(q1, y) + torch.nn.functional.mse_loss(q2, y)
        opt_c.zero_grad(); qf.backward(); opt_c.step()
        pi, logp, _ = pol.sample([img, pstate])
        q1p, q2p = crt([img, pstate, pi])
        pl = (alpha * logp - torch.min(q1p, q2p)).mean()
        opt_p.zero_grad(); opt_c.zero_grad(); pl.backward(); opt_p.step()
        soft_update(tgt, crt, tau)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(B / dt, 1), "encoder_passes": "5 fwd + 3 bwd per frame",
            "note": "transformer actor + transformer critic, DRL.py:390-432 arithmetic"}


def small_batch(dgvit_amd, synthetic, dev):
    """Secondary numbers in the reference's own regime (BASELINE config 1 / SURVEY 7 hard part 8): one 128x160 frame through the
    shipped actor's sample() (SAC.choose_action, DRL.py:170-185), and one SAC learn() step of the shipped configuration
    (config.yaml: GoT actor L4/H4/D64 + CNN critic, batch 32; DRL.py:373-437), each replayed as ONE captured HIP graph."""
    import copy
    from dgvit_amd.optim import FlatAdam, soft_update
    torch.manual_seed(3407)
    pol = dgvit_amd.GoTPolicy(2, 2, 4, 4, 64).to(dev)
    img1, ps1, _, _ = (t.to(dev) for t in synthetic.make_inputs((128, 160), 1, 0))

    def one_frame():
        with torch.no_grad():
            return pol.sample([img1, ps1])

    def timed(fn, n):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    eager = timed(one_frame, 100)
    graph = timed(dgvit_amd.GraphedStep(one_frame, warmup=3), 300)
    B = 32
    crt = dgvit_amd.QNetwork(2, 2).to(dev)
    tgt = copy.deepcopy(crt)
    op, oc = FlatAdam([pol], lr=1e-3, capturable=True), FlatAdam([crt], lr=1e-3, capturable=True)
    img, ps, act, _ = (t.to(dev) for t in synthetic.make_inputs((128, 160), B, 1))
    nimg, nps, _, _ = (t.to(dev) for t in synthetic.make_inputs((128, 160), B, 2))
    rew = torch.randn(B, 1, device=dev)

    def learn():
        with torch.no_grad():
            na, nlogp, _ = pol.sample([nimg, nps])
            q1n, q2n = tgt([nimg, nps, na])
            y = rew + 0.99 * (torch.min(q1n, q2n) - 0.2 * nlogp)
        q1, q2 = crt([img, ps, act])
        qf = torch.nn.functional.mse_loss(q1, y) + torch.nn.functional.mse_loss(q2, y)
        oc.zero_grad(); qf.backward(); oc.step()
        pi, logp, _ = pol.sample([img, ps])
        q1p, q2p = crt([img, ps, pi])
        pl = (0.2 * logp - torch.min(q1p, q2p)).mean()
        op.zero_grad(); oc.zero_grad(); pl.backward(); op.step()
        soft_update(tgt, crt, 0.005)
    learn_eager = timed(learn, 20)
    learn_graph = timed(dgvit_amd.GraphedStep(learn, warmup=2), 50)
    return {"single_frame_sample_ms": {"eager": round(eager * 1e3, 4), "hip_graph": round(graph * 1e3, 4)},
            "shipped_learn_step_ms": {"eager": round(learn_eager * 1e3, 3), "hip_graph": round(learn_graph * 1e3, 3), "batch": B},
            "note": "shipped configuration of the reference (config.yaml:5,11,58-63): GoT actor L4/H4/D64 on 128x160 frames, CNN critic; "
                    "round 1: 0.41 ms per graphed sample(), 3.66 ms per graphed learn() step"}


def c5_bf16(dgvit_amd, lib, _lib, dev, batch=440, steps=10):
    """Secondary number: BASELINE config 5 (224x224 depth frames, 12-layer ViT-Base variant with goal token, bf16 storage /
    fp32 accumulate), forward only, one GPU.  FLOPs per frame: SURVEY 8(d) dense figure (34.972 GFLOP).  The config leaves the batch
    free ("B chosen to fill the GPU"): 440 frames = 86 680 token rows = 339 row panels of 256, so the 256 x 256 tile counts of the
    layer GEMMs (x3, x9, x12 column panels) fall just under whole multiples of the 256 CUs; at B = 256 the N = 768 GEMMs run 591 tiles
    = 2.31 rounds (DESIGN 3.5 holds both)."""
    import synthetic
    torch.manual_seed(5)
    m = dgvit_amd.GoT(image_size=224, patch_size=16, num_classes=2, dim=768, depth=12, heads=12, mlp_dim=3072, channels=1)
    m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
    g = torch.Generator(device="cpu").manual_seed(5)
    img, goal = torch.rand(batch, 224, 224, generator=g).to(dev), torch.randn(batch, 768, generator=g).to(dev)
    with torch.no_grad():
        for _ in range(3):
            m(img, goal)
        torch.cuda.synchronize()
        lib.dgvit_profile_sampling(7)          # events around every 7th launch of a kind (7 and 50 GEMM launches per pass: coprime)
        lib.dgvit_profile_start(4096)
        t0 = time.perf_counter()
        for _ in range(steps):
            m(img, goal)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    kinds = _lib.PROFILE_KINDS
    ms, work, cnt = (ctypes.c_double * kinds)(), (ctypes.c_double * kinds)(), (ctypes.c_longlong * kinds)()
    lib.dgvit_profile_stop(ms, work, cnt)
    work_all, cnt_all = (ctypes.c_double * kinds)(), (ctypes.c_longlong * kinds)()
    lib.dgvit_profile_totals(work_all, cnt_all)
    lib.dgvit_profile_sampling(1)

    def per_step(k):   # sampled average launch duration x launches of that kind per pass
        return ms[k] / max(1, cnt[k]) * cnt_all[k] / steps
    fwd = synthetic.fwd_flops_per_frame((224, 224), (16, 16), 768, 12, 12, mlp_dim=3072)
    gemm_tf = (work[0] / 1e12) / (ms[0] / 1e3) if ms[0] > 0 else 0.0
    # forward + backward in train mode (dense last block, activations kept, fp32 master gradients), no optimiser step
    m.train()
    tgt = torch.randn(batch, 768, generator=g).to(dev)

    def fb():
        for p_ in m.parameters():
            p_.grad = None
        ((m(img, goal) - tgt) ** 2).mean().backward()

    for _ in range(2):
        fb()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fb()
    torch.cuda.synchronize()
    dtb = (time.perf_counter() - t0) / 5
    train = {"frames_per_s": round(batch / dtb, 1), "ms_per_step": round(dtb * 1e3, 3), "tflops_dense": round(batch / dtb * 3 * fwd / 1e12, 1),
             "frac_of_bf16_peak": round(batch / dtb * 3 * fwd / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)}
    # bytes that left the L2s per stream-GEMM launch (tools/collect_profiles_c5.sh; FETCH_SIZE x2 + WRITE_SIZE, own --pmc passes): quoted from
    # the newest committed profile, and only while the GEMM kernel sources are the ones it was collected on (synthetic.profile_traffic)
    c5_traffic, c5_note = synthetic.profile_traffic("r*_c5_bf16_l2_miss_traffic.json", "gemm_bf16_stream_kernel")
    return {"workload": f"C5: GoT 224x224@16x16, L12 H12 D768 M3072 (N=197), forward, batch {batch}, bf16 storage / fp32 accumulate",
            "frames_per_s": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 3), "dtype": "bf16",
            "tflops_dense": round(batch / dt * fwd / 1e12, 1), "frac_of_bf16_peak": round(batch / dt * fwd / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
            "batch": batch, "roofline": {"bound": "mfma", "kernel": "gemm_bf16_stream_kernel (+ ring kernel for the patch GEMM)", "achieved": round(gemm_tf, 1), "peak": PEAK_BF16_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(gemm_tf / PEAK_BF16_MFMA_TFLOPS, 4),
                         "avg_launch_ms": round(ms[0] / max(1, cnt[0]), 5), "launches_per_step": int(cnt_all[0] // steps),
                         "launches_timed": int(cnt[0]), "traffic": c5_traffic,
                         "traffic_note": "bytes beyond the L2s per launch at batch 440 (Infinity Cache + HBM; rocprofv3 --pmc FETCH_SIZE(x2) / WRITE_SIZE passes): "
                                         + c5_note + "; algorithmic operands + outputs: 536 MB per launch on average"},
            "gemm_ms_per_step": round(per_step(0), 3), "attn_fwd_ms_per_step": round(per_step(1), 3),
            "norm_ms_per_step": round(per_step(3), 3), "fwd_bwd": train}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch N>1 with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL; must be set before the first HIP call
    if not torch.cuda.is_available():
        print("bench.py needs a ROCm GPU", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("DGVIT_BENCH_SHARE_GPU") == "1":   # rehearsal only: several ranks on one card
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("DGVIT_BENCH_BACKEND", "nccl")   # "gloo" only to rehearse several ranks on one card
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL
        else:
            dist.init_process_group(backend)

    import dgvit_amd
    from dgvit_amd.parallel import GradSync
    from dgvit_amd import _lib
    import synthetic                         # input generator + FLOP model (oracle/ is only imported by cpu_baseline())
    lib = dgvit_amd.load_library()

    B = args.batch
    torch.manual_seed(3407)                      # identical initial weights on every rank (config.yaml:7 SEED)
    model = dgvit_amd.GoTPolicy(2, 2, DEPTH, HEADS, DIM, image_size=IMAGE, patch_size=PATCH).to(dev).train()
    model.trans.set_schedule(dense_last_block=args.dense_last_block, wgrad_overlap=args.wgrad_overlap)   # per-module options (dgvit_config.flags)
    # several ranks: every transformer block's all-reduce starts from inside the backward, behind its gradient-ready event
    sync = GradSync([model], force_collective=args.force_collective, overlap=args.grad_overlap)
    sync.broadcast_parameters(0)
    from dgvit_amd.optim import FlatAdam
    opt = FlatAdam([model], lr=1e-4)            # torch.optim.Adam semantics, one HIP kernel per flat block
    img, pstate, _, _ = (t.to(dev) for t in synthetic.make_inputs(IMAGE, B, 3407 + rank))   # rank-local frames, resident in HBM
    g = torch.Generator(device="cpu").manual_seed(rank)
    tgt_mean, tgt_ls = torch.randn(B, 2, generator=g).to(dev), torch.randn(B, 2, generator=g).to(dev)
    torch.manual_seed(1000 + rank)               # decorrelate dropout masks across ranks

    exchange = world > 1 or args.force_collective
    ar_events = []        # (start, end) HIP events on the compute stream around the gradient exchange of each timed step

    def step(timed=False):
        sync.zero_grad()
        mean, log_std = model([img, pstate])
        loss = torch.nn.functional.mse_loss(mean, tgt_mean) + torch.nn.functional.mse_loss(log_std, tgt_ls)
        loss.backward()
        if timed and exchange:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sync.sync()       # (its handles' wait() makes this stream wait for RCCL's: the end event is behind the collectives)
            e1.record()
            ar_events.append((e0, e1))
        else:
            sync.sync()
        opt.step()
        return loss

    def fence():
        if world > 1 or args.force_collective:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # live kernel timing inside the timed region: HIP events around every PROFILE_STRIDE-th launch of each kind.  (An event pair
    # isolates its launch from its neighbours; around every launch that slows this step by ~7 %.  10 and 91 GEMM launches per step
    # are coprime, so every launch site is sampled over the steps.)
    lib.dgvit_profile_sampling(args.profile_stride)
    lib.dgvit_profile_start(8192)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(timed=True)
    fence()
    dt = time.perf_counter() - t0
    kinds = _lib.PROFILE_KINDS
    ms = (ctypes.c_double * kinds)()
    work = (ctypes.c_double * kinds)()
    cnt = (ctypes.c_longlong * kinds)()
    lib.dgvit_profile_stop(ms, work, cnt)
    work_all = (ctypes.c_double * kinds)()
    cnt_all = (ctypes.c_longlong * kinds)()
    lib.dgvit_profile_totals(work_all, cnt_all)
    lib.dgvit_profile_sampling(1)

    # the gradient exchange on its own (SURVEY 8(d) "Multi-GPU measurement"): device time between backward's end and the optimiser's start,
    # max over ranks; algorithm bandwidth = gradient bytes / that time, bus bandwidth = x 2 (n - 1) / n (ring all-reduce traffic per link)
    ar_ms = sum(a.elapsed_time(b) for a, b in ar_events) / max(1, len(ar_events)) if ar_events else 0.0
    tmax = torch.tensor([dt, ar_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt, ar_ms = tmax[0].item(), tmax[1].item()
    final_loss = loss.item()

    # north-star figure "MFMA roofline fraction of the DGViT forward at batch 512" (N = 1 only, outside the timed region):
    # the same batch through the same model, inference mode, no timing events
    forward_only = None
    if world == 1 and not args.no_overlap_ab:
        import synthetic as _syn
        model.eval()
        with torch.no_grad():
            for _ in range(3):
                model([img, pstate])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                model([img, pstate])
            torch.cuda.synchronize()
            dtf = (time.perf_counter() - t0) / args.steps
        model.train()
        ffl = _syn.fwd_flops_per_frame(IMAGE, PATCH, DIM, DEPTH, HEADS)
        ffx = _syn.fwd_flops_per_frame_executed(IMAGE, PATCH, DIM, DEPTH, HEADS, prune_last=not args.dense_last_block)
        forward_only = {"frames_per_s": round(B / dtf, 1), "ms_per_pass": round(dtf * 1e3, 3), "tflops_dense": round(B / dtf * ffl / 1e12, 2),
                        "frac_of_f32_mfma_peak": round(B / dtf * ffl / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                        "tflops_executed": round(B / dtf * ffx / 1e12, 2),
                        "frac_of_f32_mfma_peak_executed": round(B / dtf * ffx / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                        "note": "GoTPolicy forward (eval, no_grad), B=512 84x84; dense FLOPs of SURVEY 8(d); target of the north star: >= 0.5"}

    # A/B outside the timed region (N = 1 only): the same step with each layer's weight-gradient GEMMs on the library's helper
    # stream.  Not the headline: concurrent kernels stretch each other's durations, so no per-kernel roofline can be quoted for it.
    overlap_ab = None
    if world == 1 and not args.wgrad_overlap and not args.no_overlap_ab:
        model.trans.set_schedule(dense_last_block=args.dense_last_block, wgrad_overlap=True)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dto = (time.perf_counter() - t0) / args.steps
        model.trans.set_schedule(dense_last_block=args.dense_last_block, wgrad_overlap=False)
        overlap_ab = {"frames_per_s": round(B / dto, 1), "ms_per_step": round(dto * 1e3, 3),
                      "note": "same step, weight-gradient GEMMs on a helper stream beside the data-gradient chain (DESIGN 3.3); not the headline"}

    if rank == 0:
        frames = B * world * args.steps
        fps = frames / dt
        fwd = synthetic.fwd_flops_per_frame(IMAGE, PATCH, DIM, DEPTH, HEADS)
        fwd_exec = synthetic.fwd_flops_per_frame_executed(IMAGE, PATCH, DIM, DEPTH, HEADS, prune_last=not args.dense_last_block)
        gemm_tflops = (work[0] / 1e12) / (ms[0] / 1e3) if ms[0] > 0 else 0.0

        def per_step(k):   # sampled average launch duration x launches of that kind per step
            return ms[k] / max(1, cnt[k]) * cnt_all[k] / args.steps

        # HBM bytes per GEMM launch from the rocprofv3 PMC passes of this same command (tools/pmc_traffic.py): the newest committed profile,
        # null when the GEMM kernel sources have changed since it was collected (a stale file must not ride along with fresh timings)
        traffic, traffic_note = synthetic.profile_traffic("r*_hbm_traffic.json", "gemm_f32_kernel")
        out = {
            "metric": "depth frames/sec through DGViT fwd+bwd, batch 512x84x84",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3: GoTPolicy DGViT-small (84x84@12x12, L6 H8 D256 M2048, N=50 tokens) actor fwd+bwd, "
                                   "train mode (emb dropout 0.1), MSE-to-random-target loss, grad all-reduce + Adam step (fused flat-buffer HIP Adam)"
                                   + (" [one-rank RCCL group, all-reduce forced]" if args.force_collective else ""),
                       "frames_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "gflop_per_frame_fwd_bwd": round(3 * fwd / 1e9, 4),
                       "last_block": "dense" if args.dense_last_block else "token-0 rows only (identical results; FLOPs counted dense)",
                       "wgrad_overlap": bool(args.wgrad_overlap),
                       "grad_allreduce": ("none (one rank)" if not (world > 1 or args.force_collective) else
                                          "after the backward" if not args.grad_overlap else
                                          f"per transformer block from inside the backward ({sync.early_launches // max(1, args.steps + args.warmup)} early all-reduces per step), rest after it")},
            "roofline": {"bound": "mfma", "achieved": round(gemm_tflops, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(gemm_tflops / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "traffic_note": "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE(x2)/WRITE_SIZE passes: " + traffic_note,
                         "kernel": "gemm_f32_kernel (all instantiations: NT fwd, NN dgrad, TN wgrad)",
                         "launches_per_step": int(cnt_all[0] // max(1, args.steps)),
                         "avg_launch_ms": round(ms[0] / max(1, cnt[0]), 5),
                         "avg_launch_gflop": round(work[0] / max(1, cnt[0]) / 1e9, 4),
                         "launches_timed": int(cnt[0]), "launches_in_region": int(cnt_all[0]),
                         "timing": f"HIP events around every {args.profile_stride}-th launch of the kernel inside the timed region"},
            "end_to_end": {"tflops": round(fps * 3 * fwd / 1e12, 2), "frac_of_peak": round(fps * 3 * fwd / 1e12 / world / PEAK_F32_MFMA_TFLOPS, 4),
                           "tflops_executed": round(fps * 3 * fwd_exec / 1e12, 2),
                           "frac_of_peak_executed": round(fps * 3 * fwd_exec / 1e12 / world / PEAK_F32_MFMA_TFLOPS, 4),
                           "flops_note": "tflops credits the DENSE model FLOPs of SURVEY 8(d) (the token-0-only last block skips 13.7 % of them); "
                                         "tflops_executed counts only the FLOPs the schedule runs",
                           "gemm_ms_per_step": round(per_step(0), 3), "attn_fwd_ms_per_step": round(per_step(1), 3),
                           "attn_bwd_ms_per_step": round(per_step(2), 3), "final_loss": round(final_loss, 5),
                           "per_step_note": "isolated launch durations (sampled) x launches per step; back-to-back launches overlap "
                                            "at their edges, so these can add up to more than ms_per_step"},
        }
        if exchange:
            nbytes = 4 * sync.grad_numel()
            algbw = nbytes / (ar_ms * 1e-3) / 1e9 if ar_ms > 0 else None
            out["allreduce"] = {"ms_per_step": round(ar_ms, 4), "bytes": nbytes, "algbw_GBps": None if algbw is None else round(algbw, 1),
                                "busbw_GBps": None if algbw is None else round(algbw * 2 * (world - 1) / world, 1),
                                "share_of_step": round(ar_ms / (dt / args.steps * 1e3), 4),
                                "note": ("device time of GradSync.sync() on the compute stream (HIP events), max over ranks; with --grad-overlap only the part "
                                         "that was not hidden under the backward" if args.grad_overlap else
                                         "device time of GradSync.sync() on the compute stream (HIP events, one exchange after the backward), max over ranks")
                                        + "; busbw = algbw x 2 (n - 1) / n"}
        if forward_only:
            out["forward_only"] = forward_only
        if overlap_ab:
            out["wgrad_overlap_ab"] = overlap_ab
        if world == 1 and not args.no_sac_step:
            out["sac_step"] = sac_step(dgvit_amd, synthetic, B, dev)
        if world == 1 and not args.no_small_batch:
            out["small_batch"] = small_batch(dgvit_amd, synthetic, dev)
        if world == 1 and not args.no_c5:
            out["c5_bf16"] = c5_bf16(dgvit_amd, lib, _lib, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch)
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
