#!/usr/bin/env python3
"""Benchmark of the GraspPose-Opt hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts N ranks itself, one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W           (same ranks under torchrun; --gpus must equal the world size)
    python bench.py --gpus 8 --batch 2048                  (BASELINE config 3: B = 16,384 = 8 x 2,048 candidates)

One "step" = one fitting iteration of stage 3 (every loss term active) over B candidates on each GPU:
6D->aa + VPoser decode, SMPL-X forward (full 10,475-vertex mesh), hand vertex normals, two bidirectional
nearest-neighbour searches against 2,048 object points, signed distances + contact / collision losses,
backward to the optimiser variables, Adam step -- `saga_fit_step`, one kernel sequence with no host work in
between (`--autograd` runs the same iteration as PyTorch autograd over the drop-in operators + torch.optim.Adam).  Inputs are synthetic (random-init
shape-compatible body model, ellipsoid object clouds, seeded candidates) and resident in HBM before the timed
region.  Candidates shard across ranks with no data-path collective ("weak" scaling: B per GPU is fixed);
the end-of-fit all-gather for best-of-N runs once, after the timed region.

Rank 0 prints ONE JSON line; see DESIGN.md 'Measurement' for the roofline arithmetic.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import hashlib

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC (before the HIP runtime starts): RCCL across processes needs it here

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3     # MI355X dense FP32 (vector = matrix), /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0       # HBM3E spec peak (same guide; ~6.3 TB/s achievable)
V, K_ALG = 10475, 506       # algorithmic K of the blend contraction: 20 shape + 486 pose-feature columns (padded to 512)
PMC_SUMMARY = os.path.join(ROOT, 'profiles', 'current_pmc_summary.json')
GEMM_ENV_FLAGS = ('SAGA_GEMM_LEPI', 'SAGA_GEMM_XCD', 'SAGA_GEMM_XT', 'SAGA_GEMM_XT_BWD')


def stage_rooflines(prof, steps, B, n_obj, V=10475):
    """Per-kernel achieved rate against the roofline that bounds it (SURVEY.md 8d figures x B / live HIP-event time)."""
    ms = lambda k: prof[k][0] / max(prof[k][1], 1) if k in prof and prof[k][1] else None
    out = {}

    def add(name, key, bound, work, unit, peak):
        t = ms(key)
        if t:
            rate = work / (t * 1e-3) / (1e12 if unit == 'TFLOP/s' else 1e9)
            out[name] = {'bound': bound, 'ms': round(t, 4), 'achieved': round(rate, 2), 'unit': unit, 'peak': peak,
                         'frac': round(rate / peak, 4)}
    gemm = 2.0 * B * 3 * V * K_ALG
    add('blend_gemm_fwd', 'blend_gemm_fwd', 'mfma', gemm, 'TFLOP/s', PEAK_F32_TFLOPS)
    add('blend_gemm_bwd', 'blend_gemm_bwd', 'mfma', gemm, 'TFLOP/s', PEAK_F32_TFLOPS)
    vb = 3 * V * 4.0                                           # bytes of one [V,3] row
    add('skin_fwd', 'skin_fwd', 'hbm', B * 2 * vb, 'GB/s', PEAK_HBM_GBS)                 # read v_posed, write vertices
    add('skin_bwd', 'skin_bwd', 'hbm', B * 3 * vb, 'GB/s', PEAK_HBM_GBS)                 # read g_vertices + v_posed, write g_v_posed
    add('grasp_loss_bwd', 'grasp_loss_bwd', 'hbm', B * vb, 'GB/s', PEAK_HBM_GBS)         # dense g_vertices written once
    add('pose_chain_fwd', 'pose_fwd', 'hbm', B * (512 + 55 * 36 + 165) * 4.0, 'GB/s', PEAK_HBM_GBS)
    # NN: (143 + 778) x O point pairs per candidate, 8 FLOP each (3 sub, 1 mul, 2 fma), one pass serving both directions -- the
    # SURVEY 8d figure, priced against the FP32 vector peak.  (The kernel evaluates both directions, 3.48 M pairs per candidate;
    # its approximate pass runs on the matrix cores, the exact pass and the bookkeeping on the vector units.)
    add('grasp_loss_fwd_nn', 'grasp_loss_fwd', 'valu', B * (143 + 778) * n_obj * 8.0, 'TFLOP/s', PEAK_F32_TFLOPS)
    return out


def source_sha16(name):
    """sha256[:16] of a kernel source file under saga_amd/csrc (what ties a PMC summary to the code it profiled)."""
    try:
        return hashlib.sha256(open(os.path.join(ROOT, 'saga_amd', 'csrc', name), 'rb').read()).hexdigest()[:16]
    except OSError:
        return None


def pmc_traffic(B, n_obj, V):
    """(HBM bytes per launch of the dominant kernel, note).  The PMC counters cannot be collected inside a timed run, so the
    figure comes from the committed rocprofv3 --pmc summary of THIS command (scripts/profile_round.sh: separate passes,
    FETCH_SIZE doubled as the guide prescribes for gfx950) -- and only when that summary describes this run: same blend GEMM
    source (sha256 of blend_gemm.hip recorded in the summary's _meta), same B / O / V, same GEMM environment switches.
    Otherwise traffic is None and the note says what differed."""
    try:
        d = json.load(open(PMC_SUMMARY))
    except (OSError, ValueError):
        return None, 'no PMC summary at %s' % os.path.relpath(PMC_SUMMARY, ROOT)
    meta = d.get('_meta') or {}
    here = {'blend_gemm.hip': source_sha16('blend_gemm.hip'), 'B': B, 'n_obj': n_obj, 'V': V,
            'env': {k: os.environ.get(k, '') for k in GEMM_ENV_FLAGS}}
    there = {'blend_gemm.hip': (meta.get('src_sha16') or {}).get('blend_gemm.hip'), 'B': meta.get('B'), 'n_obj': meta.get('n_obj'),
             'V': meta.get('V'), 'env': {k: (meta.get('env') or {}).get(k, '') for k in GEMM_ENV_FLAGS}}
    diff = [k for k in here if here[k] != there[k]]
    if diff:
        return None, 'PMC summary %s does not describe this run (differs in: %s)' % (os.path.relpath(PMC_SUMMARY, ROOT), ', '.join(diff))
    g = [v for k, v in d.items() if k.startswith(('gemm_nt_f32_kernel', 'blend_fwd_xt_kernel', 'blend_bwd_xt_kernel')) and v.get('hbm_bytes') and v.get('mean_us', 0) > 500]
    if not g:
        return None, 'PMC summary holds no blend GEMM launches'
    return round(sum(x['hbm_bytes'] for x in g) / len(g)), ('offline: %s (git %s), HBM-side bytes per launch, mean of the forward and backward '
                                                           'launches (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc)'
                                                           % (os.path.relpath(PMC_SUMMARY, ROOT), meta.get('git_head', '?')))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None,
                    help='GPUs (= ranks, one process each) on this node; default: WORLD_SIZE when launched by torchrun, else 1')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=4096, help='candidates per GPU')
    ap.add_argument('--obj-points', type=int, default=2048)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--autograd', action='store_true', help='PyTorch autograd + torch.optim.Adam instead of saga_fit_step')
    ap.add_argument('--loss-vertices-only', action='store_true',
                    help='SEPARATELY REPORTED variant (SURVEY 8d, V_sel): the body model evaluates only the ~1,100 vertices the loss reads')
    ap.add_argument('--sparse-backward', action='store_true',
                    help='SEPARATELY REPORTED variant: saga_fit_step_sparse -- full-mesh forward, backward over the ~1,050 vertices '
                         'the loss reads (the dense vertex gradient is zero everywhere else)')
    ap.add_argument('--local-weights', action='store_true',
                    help='synthetic skinning weights with mesh-like locality (runs of vertices share their joints); reported separately')
    ap.add_argument('--mesh-faces', action='store_true',
                    help='synthetic template + faces of a mesh-like tube (face corners a few ids apart) instead of random points joined by '
                         'random triangles; reported separately')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip config.secondary (the other BASELINE configs, measured live after the headline region: ~40 s)')
    ap.add_argument('--cpu-seconds', type=float, default=7.0,
                    help='CPU baseline: time budget of EACH of its two legs (B=1 reference semantics, B=64 batched)')
    return ap.parse_args()


def build_problem(B, n_obj, rank, model_root, device, loss_vertices_only=False):
    from saga_amd.fitting import FittingOP
    from saga_amd.synthetic import make_candidate_params, make_grasp_targets, make_object_cloud
    from saga_amd.vposer import VPoserDecoder
    vp = VPoserDecoder(seed=0)
    op = FittingOP({'batch_size': B, 'device': device, 'body_model_path': model_root, 'vposer': vp, 'genders': ('male',),
                    'loss_vertices_only': loss_vertices_only})
    op.bm = op.bm_male
    op.set_variables(make_candidate_params(B, seed=1234 + 7919 * rank, scale=0.3))
    vo, no, _, _ = make_object_cloud(B, n_points=n_obj, seed=rank)
    mk, mc, oc = make_grasp_targets(B, vo, seed=4815 + rank)
    cu = lambda a: torch.from_numpy(a).to(device)
    data = dict(markers=cu(mk), mc=cu(mc), oc=cu(oc), vo=cu(vo), no=cu(no))
    data['problem'] = op.make_problem(data['markers'], data['vo'], data['no'], data['oc'], data['mc'])
    return op, vp, data


def ring_size(op):
    """Vertices the hand's vertex normals read (every corner of a face touching a hand vertex) on the model in use."""
    return int(op.tables.struct_for_mesh(op.bm.faces, op.bm.num_verts).n_ring)


def step(op, d, optimizer=None):
    if optimizer is None:                      # native: whole iteration (incl. Adam) inside libsaga_hip.so
        return op.native_step(2, d['problem'])[0]
    optimizer.zero_grad(set_to_none=True)
    loss, *_ = op.calc_loss_contact_map(d['markers'], d['vo'], d['no'], d['oc'], d['mc'], 'male', None, 2, 1.0)
    loss.sum().backward()
    optimizer.step()
    return loss


def cpu_baseline(model_root, vp, n_obj, batch, seconds):
    """The oracle (CPU restatement of the reference iteration, PyTorch eager FP32) timed on the host cores: as many
    fwd+bwd iterations at ``batch`` candidates as fit ``seconds`` (at least 2).  -> (candidate-iterations/s, threads, s, iterations)"""
    from oracle import loss_ref, smplx_ref
    from saga_amd.fitting import load_grasp_indices
    from saga_amd.synthetic import make_candidate_params, make_grasp_targets, make_object_cloud
    threads = torch.get_num_threads()
    data = dict(np.load(os.path.join(model_root, 'smplx', 'SMPLX_MALE.npz')))
    bm = smplx_ref.SMPLXRef(data, num_pca_comps=24, batch_size=batch, dtype=torch.float32)
    idx_np = load_grasp_indices()
    idx = {'markers_143': torch.from_numpy(idx_np['markers_143']), 'foot': torch.from_numpy(idx_np['foot']),
           'rhand': torch.from_numpy(idx_np['rhand']), 'v_weights': torch.from_numpy(idx_np['rhand_weight'])}
    var = {k: torch.from_numpy(v).requires_grad_(True) for k, v in make_candidate_params(batch, seed=1234, scale=0.3).items()}
    vo, no, _, _ = make_object_cloud(batch, n_points=n_obj, seed=0)
    mk, mc, oc = make_grasp_targets(batch, vo, seed=4815)
    t = torch.from_numpy
    w = vp.weights_dict(torch.float32)

    def one():
        for v in var.values():
            v.grad = None
        total, *_ = loss_ref.loss_batched(bm, var, w, idx, t(mk), t(vo), t(no), t(oc), t(mc), 2)
        total.sum().backward()

    one()                                   # warm-up (allocator, thread pool)
    iters, t0 = 0, time.perf_counter()
    while iters < 2 or time.perf_counter() - t0 < seconds:
        one()
        iters += 1
    dt = time.perf_counter() - t0
    return batch * iters / dt, threads, dt, iters


def secondary_measurements(model_root, B, n_obj, device):
    """The BASELINE configs other than the headline, each measured LIVE in this process after the headline's timed region (a few
    iterations each; VERDICT r2: numbers that exist only as builder-run files count as unmeasured).  -> dict for config.secondary."""
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import workloads as wl
    import saga_amd.chamfer_distance as chd
    sec, notes = {}, {}

    def guard(name, fn):
        t0 = time.perf_counter()
        try:
            fn()
        except Exception as e:                      # a secondary line must never take the headline down with it
            sec[name] = None
            notes[name] = 'failed: %s: %s' % (type(e).__name__, str(e)[:200])
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        notes.setdefault('wall_s', {})[name] = round(time.perf_counter() - t0, 1)

    def sparse():
        op, _, d = build_problem(B, n_obj, 0, model_root, str(device))
        op.sparse_backward = True
        sec['sparse_backward_ms'] = round(wl.timed(lambda: step(op, d), 20, 5), 4)
        op.close()
    guard('sparse_backward_ms', sparse)

    def b1024():
        op, _, d = build_problem(1024, n_obj, 0, model_root, str(device))
        sec['batch1024_ms'] = round(wl.timed(lambda: step(op, d), 20, 5), 4)
        op.close()
    guard('batch1024_ms', b1024)

    def motion():
        op, run = wl.motion_stage2(model_root, 256, 61, 30, str(device))
        run(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(20)                                   # (one fitting_stage2 call of 20 iterations: its per-call set-up is amortised as in a fit)
        torch.cuda.synchronize()
        sec['motion_stage2_256x61_ms'] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    guard('motion_stage2_256x61_ms', motion)

    def chamfer():
        x, y = wl.dense_chamfer_clouds(512, 10475, 100000, str(device))
        sec['dense_chamfer_10475x100k_B512_ms'] = round(wl.timed(lambda: chd.ChamferDistance(x, y), 3, 1), 2)
    guard('dense_chamfer_10475x100k_B512_ms', chamfer)

    def fit():
        seconds, op, _ = wl.whole_fit(model_root, B, (300, 400, 500), str(device))
        sec['whole_fit_1200_iters_s'] = round(seconds, 3)
        op.close()
    guard('whole_fit_1200_iters_s', fit)
    sec['what'] = ('measured live in this process after the headline region, inputs resident in HBM: sparse_backward_ms = the headline '
                   'workload through saga_fit_step_sparse (backward over the loss vertices only); batch1024_ms = BASELINE config 2 '
                   '(B = 1024); motion_stage2_256x61_ms = config 4, one GraspMotion-Opt stage-2 iteration over 256 x 61 = 15,616 bodies; '
                   'dense_chamfer_10475x100k_B512_ms = config 5, ChamferDistance both directions, 5.36e11 point pairs each; '
                   'whole_fit_1200_iters_s = a whole 300 + 400 + 500-iteration fit of B = %d candidates (sustained: %s ms per iteration)'
                   % (B, 'n/a' if not sec.get('whole_fit_1200_iters_s') else round(sec['whole_fit_1200_iters_s'] / 1.2, 3)))
    sec['notes'] = notes
    return sec


def rccl_gather_at_world_1(loss, params, B, device):
    """At N = 1 no process group exists, so the end-of-fit exchange would never touch RCCL in a driver-run bench.  Start a 1-rank
    `nccl` (= RCCL) group after the timed region and run the same all-gathers through it (SAGA_BENCH_RCCL=0 skips this).
    -> (losses, params, note)"""
    from saga_amd.distributed import gather_candidates
    if os.environ.get('SAGA_BENCH_RCCL', '1') == '0':
        return gather_candidates(loss, params, n_total=B) + ('skipped (SAGA_BENCH_RCCL=0): single process, no collective',)
    if dist.is_initialized():       # a 1-rank launch by torchrun: main() has set the group up over the launcher's store
        gl, gp = gather_candidates(loss, params, n_total=B, force_collective=True)
        return gl, gp, 'RCCL all_gather_into_tensor executed in the 1-rank nccl group of the launcher'
    try:
        from saga_amd.launch import free_port
        t0 = time.perf_counter()
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % free_port(), rank=0, world_size=1, device_id=device)
        gl, gp = gather_candidates(loss, params, n_total=B, force_collective=True)
        torch.cuda.synchronize()
        dist.destroy_process_group()
        return gl, gp, 'RCCL all_gather_into_tensor executed in a 1-rank nccl group (%.2f s incl. group set-up)' % (time.perf_counter() - t0)
    except Exception as e:
        return gather_candidates(loss, params, n_total=B) + ('RCCL group failed (%s: %s): gathered locally' % (type(e).__name__, str(e)[:160]),)


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and (args.gpus or 1) > 1:
        # plain `python bench.py --gpus N`: start the N ranks here, before anything touches a GPU in this process
        from saga_amd.launch import launch_ranks
        try:
            sys.exit(launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
        except RuntimeError as e:
            sys.exit('bench.py: %s' % e)
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if args.gpus is not None and args.gpus != world:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d: launch as many ranks as GPUs (or drop --gpus under torchrun)' % (args.gpus, world))
    if torch.cuda.device_count() <= local:
        sys.exit('bench.py: rank %d needs GPU %d, but %d GPU(s) are visible' % (rank, local, torch.cuda.device_count()))
    # Under a launcher (torchrun sets TORCHELASTIC_RUN_ID) the group comes from the launcher's store at any world size, 1 included: a
    # tcp:// rendezvous of our own would be taken for a CLIENT of the agent's store there and wait for a server that does not exist.
    if world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
    assert torch.cuda.is_available(), 'bench.py needs a ROCm GPU'
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    from saga_amd import _lib
    from saga_amd.distributed import best_of_n, gather_candidates
    from saga_amd.synthetic import write_body_model

    B, n_obj = args.batch, args.obj_points
    tmp = tempfile.TemporaryDirectory()
    write_body_model(tmp.name, gender='male', seed=7, local_weights=args.local_weights, mesh_faces=args.mesh_faces)
    op, vp, d = build_problem(B, n_obj, rank, tmp.name, str(device), args.loss_vertices_only)
    op.sparse_backward = bool(args.sparse_backward) and not args.autograd
    V_eval = op.bm.num_verts
    optimizer = op.optimizer_s3 if args.autograd else None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(op, d, optimizer)
    barrier()
    # timed region: HIP events bracket the launches of the roofline kernel only (the blend GEMMs); the per-kernel
    # breakdown of every other kernel comes from a second, untimed pass of the same steps
    _lib.profile_enable(2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(op, d, optimizer)
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.profile_enable(False)
    prof_gemm = _lib.profile_read()
    _lib.profile_enable(1)
    for _ in range(args.steps):
        step(op, d, optimizer)
    barrier()
    _lib.profile_enable(False)
    prof = _lib.profile_read()

    t_max = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())

    # end-of-fit exchange (untimed): all-gather (loss, parameters) of every candidate, best-of-N per group of 8
    with torch.no_grad():
        params = {k: v.detach().reshape(B, -1) for k, v in op.body_params().items()}
    if world == 1:
        gl, gp, gather_note = rccl_gather_at_world_1(loss.detach(), params, B, device)
    else:
        gl, gp = gather_candidates(loss.detach(), params, n_total=B * world)
        gather_note = 'RCCL all_gather_into_tensor over %d ranks' % world
    _, best_loss, _ = best_of_n(gl, gp, group_size=8)
    default_run = not (args.loss_vertices_only or args.sparse_backward or args.local_weights or args.autograd or args.mesh_faces)
    secondary = None
    if rank == 0 and world == 1 and default_run and not args.no_secondary:
        secondary = secondary_measurements(tmp.name, B, n_obj, device)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        iters_per_s = world * args.steps / elapsed              # B-candidate iterations over all GPUs
        # dominant kernel: the FP32-MFMA blend contraction (forward + backward launches are the same kernel template)
        gemm_ms = prof_gemm['blend_gemm_fwd'][0] + prof_gemm['blend_gemm_bwd'][0]      # live, inside the timed region
        gemm_n = prof_gemm['blend_gemm_fwd'][1] + prof_gemm['blend_gemm_bwd'][1]
        if args.sparse_backward:       # the backward contraction runs over the loss vertices only: price the forward launches
            gemm_ms, gemm_n = prof_gemm['blend_gemm_fwd']
        flop_per_launch = 2.0 * B * (3 * V_eval) * K_ALG       # algorithmic: [B,506] x [506,31425], fwd == bwd
        achieved = flop_per_launch / (gemm_ms / max(gemm_n, 1) * 1e-3) / 1e12 if gemm_n else None
        kernels = {k: {'ms_per_step': round(ms / args.steps, 4), 'launches_per_step': n / args.steps}
                   for k, (ms, n) in prof.items() if n}
        out = {
            'metric': 'GraspPose-Opt fwd+bwd iters/sec at B=4096, 2048 obj pts (1/2/4/8 GPU)',
            'value': round(iters_per_s, 4),
            'unit': 'iterations/s (each = one fwd+bwd+Adam iteration over B=%d candidates per GPU, summed over GPUs)' % B,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'GraspPose-Opt stage-3 iteration: SMPL-X fwd+bwd (%s) + hand normals + '
                                   '2 bidirectional NN vs %d object points + contact/collision losses + Adam'
                                   % ('V=10475, full mesh' if not args.loss_vertices_only else
                                      'VARIANT: only the %d loss vertices of 10475 are evaluated' % V_eval, n_obj),
                       'skinning_weights': ('VARIANT: 4 non-zeros per vertex, runs of vertices share their joints (mesh-like)' if args.local_weights
                                            else '4 non-zeros per vertex at random joints (SURVEY 8d synthetic model)'),
                       'faces': ('VARIANT: mesh-like tube (face corners a few ids apart; hand ring %d vertices)' if args.mesh_faces else
                                 'random triangles (SURVEY 8d synthetic model; hand ring %d vertices)') % ring_size(op),
                       'driver': ('torch autograd + torch.optim.Adam' if args.autograd else
                                  'VARIANT: saga_fit_step_sparse (backward over the %d loss vertices only)' % op.tables.n_unique
                                  if args.sparse_backward else 'saga_fit_step (one kernel sequence)'),
                       'candidates_per_gpu': B, 'candidates_total': B * world, 'object_points': n_obj,
                       'candidate_iterations_per_s': round(iters_per_s * B, 1),
                       'parallelism': 'candidates sharded, %d rank(s), no data-path collective' % world,
                       'best_of_8_mean_loss': float(best_loss.mean()), 'end_of_fit_gather': gather_note, 'kernels': kernels,
                       'stages': stage_rooflines(prof, args.steps, B, n_obj, V_eval)},
            'roofline': {'bound': 'mfma', 'kernel': 'blend_fwd_xt_kernel + blend_bwd_xt_kernel (blend contraction: the forward and the split-K backward launch; gemm_nt_f32_kernel at small batches)',
                         'achieved': None if achieved is None else round(achieved, 2), 'peak': PEAK_F32_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': None if achieved is None else round(achieved / PEAK_F32_TFLOPS, 4),
                         'traffic': None, 'traffic_source': None,
                         'algorithmic_bytes': int(B * 3 * V_eval * 4 + B * 512 * 4 + 3 * V_eval * 512 * 4)},
        }
        if args.loss_vertices_only or args.sparse_backward or args.local_weights or args.autograd or args.mesh_faces:
            out['roofline']['traffic_source'] = 'not profiled for this variant'
        else:
            out['roofline']['traffic'], out['roofline']['traffic_source'] = pmc_traffic(B, n_obj, V_eval)
        if args.sparse_backward:        # the dense algorithmic figures do not describe this variant's backward kernels
            for k in ('blend_gemm_bwd', 'skin_bwd', 'grasp_loss_bwd'):
                out['config']['stages'].pop(k, None)
        if secondary is not None:
            out['config']['secondary'] = secondary
        if default_run:
            out['config']['variants_reported_separately'] = ('bench.py --sparse-backward, --loss-vertices-only, --mesh-faces / --local-weights '
                                                             '(mesh-like synthetic model), --autograd; never the headline')
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is timed at N = 1 only (host cores are shared by the ranks)
            # the oracle at B=1 (the reference's own semantics: one candidate at a time) AND at B=64 (batched, to be
            # fair to the CPU); value = the better of the two
            legs = []
            for cb in (1, 64):
                cand_per_s, threads, dt, iters = cpu_baseline(tmp.name, vp, n_obj, cb, args.cpu_seconds)
                legs.append((cand_per_s, 'B=%d: %d fwd+bwd iterations in %.1f s = %.2f candidate-iterations/s' % (cb, iters, dt, cand_per_s)))
            best = max(l[0] for l in legs)
            out['cpu_baseline'] = {
                'value': round(best / B, 6), 'unit': 'iterations/s at B=%d (extrapolated from candidate-iterations/s)' % B,
                'cores': threads, 'kind': 'port',
                'sample': 'oracle (PyTorch CPU eager FP32, %d threads) on the same workload, %d object points; %s; %s; value = the better leg'
                          % (threads, n_obj, legs[0][1], legs[1][1])}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    tmp.cleanup()


if __name__ == '__main__':
    main()
