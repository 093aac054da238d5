"""End-to-end OCR throughput on synthetic dense-text pages (BASELINE.json config 4: 1024x1024).

    python bench.py --gpus 1 --steps 40 --warmup 3
    python bench.py --gpus N ...            # starts its own N ranks (torch.distributed.run) when not already one of them
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over a batch of pages: host -> device copy of the uint8 pages
(pinned memory, a copy stream of its own, one batch ahead of the compute stream; `--resident` re-feeds a
batch that already sits in HBM instead) -> EAST-VGG16 forward (f32 MFMA) -> RBOX decode -> LANMS + NMS
-> crop/resize/pack -> CRNN + CTC greedy -> strings on the host.  Images are independent, so ranks share no data (weak scaling: N x steps
batches per job, no collective in the data path).  The batches of the timed region sit in ONE queue
all ranks take from (post-NMS word counts differ, a rank that is done early takes more), and every
rank's boxes + strings are gathered on the host of rank 0 before the clock stops; one barrier and
one MAX all-reduce (RCCL) bracket the timed region.

No trained weights exist offline: both networks run with seeded random weights of the reference
architectures (all their arithmetic is executed and timed), and the decode stage consumes the
synthetic "teacher" maps of the rendered pages instead of the random-weight network output, so that
decode / LANMS / crop / CRNN see realistic box counts (~18 000 candidates and ~200 words per page).

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (a 3x3 f32-MFMA
convolution), timed with HIP events on the launch stream inside the timed region by the library's
stage timers: `achieved` / `frac` count the flops the matrix cores EXECUTE (a Winograd kernel
executes 36/144 or 16/36 of the direct convolution's), `direct_equiv_tflops` the convolution it
computes.  `roofline_hbm` lists the bandwidth-bound stages against 8 TB/s.  `cpu_baseline` times the
reference-faithful CPU path (oracle/) on the host cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X dense f32 matrix peak (= f32 vector peak)
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X dense bf16 matrix peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=48, help='pages per step and per GPU (25 GB of activations in flight per 48 pages)')
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--words', type=int, default=220)
    ap.add_argument('--vocab', type=int, default=97, help='classes of the recogniser head (97 = english_g2, 6719 = zh_sim_g2)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="f32 = the reference's precision (the headline); bf16 = bfloat16 operands, float32 accumulation")
    ap.add_argument('--rotated', action='store_true',
                    help='opt-in path that keeps the quadrilaterals: rotated-IoU NMS + rotated crops (not the '
                         'reference behaviour; the CPU baseline is skipped)')
    ap.add_argument('--reference-defaults', action='store_true',
                    help="the reference's default arguments instead of the fast configuration: words grouped into lines and "
                         "paragraphs (combine=True), CTC beam search with 10 beams (not the headline; the CPU baseline is skipped)")
    ap.add_argument('--resident', action='store_true',
                    help='pages already in HBM when the clock starts (no host -> device copy inside the timed region)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-pages', type=int, default=4)
    ap.add_argument('--set', action='append', default=[], metavar='NAME=VALUE',
                    help='library option for ocrhip_set_option (equivalent algorithms, e.g. fuse_ctc=0, lstm_persistent=0)')
    ap.add_argument('--share-gpu', action='store_true',
                    help='rehearsal of the N-rank path on a box with fewer GPUs: rank r uses GPU r %% visible, gloo instead of '
                         'RCCL for the two bracket collectives (RCCL refuses two ranks on one device), per-step LSTM kernel '
                         '(the persistent one needs the whole GPU); the line it prints is not a scaling result')
    return ap.parse_args()


def spawn_ranks(args):
    """`--gpus N` outside torchrun: start N ranks as children of this process, which never touches a GPU itself."""
    have = torch.cuda.device_count()          # counts devices without initialising the GPU runtime
    if have < args.gpus and not (args.share_gpu and have >= 1):
        sys.stderr.write('bench.py: --gpus %d asked for, %d GPU(s) visible\n' % (args.gpus, have))
        return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def cpu_baseline(images, maps, east_params, crnn_params, n_pages):
    """Reference-faithful CPU path on a bounded sample (same pages as the GPU run)."""
    from oracle import ctc as octc, pipeline as opipe
    from ocr_amd.utils.text import ENGLISH_G2_CHARACTERS
    tok = octc.CharTokenizer(ENGLISH_G2_CHARACTERS)
    t0 = time.perf_counter()
    words = 0
    for b in range(n_pages):
        res = opipe.ocr_page(images[b], east_params, crnn_params, tok, maps_override=maps[b], trig='numpy32')
        words += len(res['texts'])
    dt = time.perf_counter() - t0
    # the same arithmetic batched the way a CPU implementation would (one EAST forward over the pages, one recogniser
    # forward per crop width): not what the reference does, reported beside it
    t1 = time.perf_counter()
    res = opipe.ocr_pages_batched(images[:n_pages], east_params, crnn_params, tok, maps_override=maps[:n_pages], trig='numpy32')
    dt2 = time.perf_counter() - t1
    words2 = sum(len(r['texts']) for r in res)
    return {'value': n_pages / dt, 'unit': 'images/s', 'cores': int(torch.get_num_threads()), 'kind': 'port',
            'words_per_s': words / dt,
            'sample': '%d page(s) %dx%d of the GPU workload through oracle/pipeline.py: batch-1 EAST (torch-CPU f32), '
                      'NumPy decode, C LANMS (1 thread), one CRNN forward per text row; %.1f s'
                      % (n_pages, images.shape[1], images.shape[2], dt),
            'batched': {'value': n_pages / dt2, 'unit': 'images/s', 'words_per_s': words2 / dt2,
                        'sample': 'the same pages, one EAST forward over all of them and one CRNN forward per crop width; %.1f s' % dt2}}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed counter passes of this same command
    (tools/pmc_traffic.py over `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`), or None."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fh:
            return json.load(fh)['kernels'][kernel]['traffic_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def stage_accounting(pipe, d_img, d_maps):
    """Data-dependent work of one batch for the bandwidth-bound stages (one untimed pass after the timed region):
    decode candidates, LANMS pair tests, crop source / output pixels, CTC rows."""
    from ocr_amd.utils import image_sizes as S
    B, H, W, _ = d_img.shape
    _, dec, kept = pipe.detect(d_img, d_maps)
    torch.cuda.synchronize()
    cand = dec['counts'].cpu().numpy().astype(np.int64)
    n_iter = kept['n_iter'].cpu().numpy().astype(np.int64) if 'n_iter' in kept else np.zeros(B, np.int64)
    counts = kept['counts'].cpu().numpy()
    boxes = kept['boxes'].cpu().numpy()
    src_px = out_elems = ctc_rows = 0
    for b in range(B):
        if pipe.rotated:
            p = S.plan_crops_rotated(boxes[b, :int(counts[b])].reshape(-1, 4, 2), H, W)
            src_px += int((p['geom'][:, 6].astype(np.int64) * p['geom'][:, 7].astype(np.int64)).sum())
        else:
            p = S.plan_crops(boxes[b, :int(counts[b])], H, W)
            r = p['rects'].astype(np.int64)
            src_px += int(((r[:, 2] - r[:, 0]) * (r[:, 3] - r[:, 1])).sum())
        out_elems += int(64 * p['out_w'].astype(np.int64).sum())
        ctc_rows += int(pipe.ctx.seq_len(p['out_w']).sum())
    return {'decode_candidates': int(cand.sum()), 'decode_map_px': B * (H // 4) * (W // 4),
            'lanms_pairs_first_iter': int((cand * cand).sum()), 'lanms_iters_max': int(n_iter.max(initial=0)),
            'lanms_pairs_dense': int((cand * cand * n_iter).sum()), 'crop_src_px': src_px, 'crop_out_elems': out_elems,
            'ctc_rows': ctc_rows}


def hbm_rooflines(stats, acct, vocab, steps):
    """Algorithmic bytes per launch of the bandwidth-bound stages / their HIP-event time, against 8 TB/s."""
    by = {s['name']: s for s in stats}
    per_launch = {
        # 24 B x map pixels read + 20 B x boxes written (16 B xyxy + 4 B score; the pipeline asks for no polygons)
        'east_decode': 24.0 * acct['decode_map_px'] + 20.0 * acct['decode_candidates'],
        # 20 B per box in and out; the work is pair tests, not bytes
        'lanms': 2 * 20.0 * acct['decode_candidates'],
        # 3 B x crop source pixels read + 4 B x 64 x Wpad written
        'crop_resize_pack': 3.0 * acct['crop_src_px'] + 4.0 * acct['crop_out_elems'],
        # 4 B x T x V logits read + 4 B x T tokens written (the path that writes logits: fuse_ctc=0, bf16)
        'ctc_greedy': 4.0 * acct['ctc_rows'] * (vocab + 1),
        # greedy decoding fused into the class layer: 8 B per row and block of 128 classes read + 4 B x T written
        'ctc_greedy_partial': 4.0 * acct['ctc_rows'] * (2 * ((vocab + 127) // 128) + 1),
    }
    pmc_names = {'east_decode': ['east_decode_flags', 'east_decode_write'], 'ctc_greedy_partial': ['ctc_greedy']}
    out = []
    for name, nbytes in per_launch.items():
        s = by.get(name)
        if not s or not s['launches']:
            continue
        avg_ms = s['total_ms'] / s['launches']
        parts = [pmc_traffic(k) for k in pmc_names.get(name, [name])]
        e = {'kernel': name, 'launches_per_step': s['launches'] / steps, 'algorithmic_bytes': nbytes, 'avg_ms': avg_ms,
             'achieved_GBs': nbytes / (avg_ms * 1e-3) / 1e9, 'frac_of_8TBs': nbytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
             'traffic': None if any(p is None for p in parts) else sum(parts)}
        if name == 'lanms':
            e['bound'] = 'latency / pair tests, not HBM'
            e['pair_tests_first_iteration'] = acct['lanms_pairs_first_iter']
            e['iterations_max'] = acct['lanms_iters_max']
            e['dense_pairs_per_s'] = acct['lanms_pairs_dense'] / (avg_ms * 1e-3)
        out.append(e)
    return out


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(spawn_ranks(args))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.share_gpu:
        local_rank %= max(1, torch.cuda.device_count())
    if local_rank >= torch.cuda.device_count():
        sys.exit('bench.py: rank with LOCAL_RANK=%d has no GPU (%d visible)' % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    from ocr_amd.dist import PageQueue, gather_results, init_from_env, max_over_ranks
    device = torch.device('cuda', local_rank)
    grp = init_from_env('gloo' if args.share_gpu else 'nccl', device)          # RCCL when launched as one of N ranks
    rank, world, dist = grp
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but the job has %d rank(s)' % (args.gpus, world))

    from ocr_amd import synth
    from ocr_amd.pipeline import OcrPipeline
    from ocr_amd.weights import random_east_params, random_crnn_params

    east_params = random_east_params(seed=1)
    crnn_params = random_crnn_params(vocab_size=args.vocab, seed=2)
    from ocr_amd._lib import BF16, F32
    pipe = OcrPipeline(east_params, crnn_params, device=local_rank, compute_dtype=BF16 if args.dtype == 'bf16' else F32,
                       rotated=args.rotated, **(dict(combine=True, method='beam_search', num_beams=10) if args.reference_defaults else {}))

    if args.share_gpu:
        pipe.ctx.set_option('lstm_persistent', 0)
    for kv in args.set:
        name, value = kv.split('=')
        pipe.ctx.set_option(name, int(value))
    B, S = args.batch, args.size
    images, maps, _ = synth.make_pages(1234 + 4 + 100 * rank, B, S, S, args.words)
    d_img = torch.from_numpy(images).cuda()
    d_maps = torch.from_numpy(maps).cuda()
    # Host -> device inside the clock: the pages of every batch are copied from pinned host memory on a copy stream, one
    # batch ahead of the compute stream, into a ring of three device buffers (a batch's pages are read by its detector
    # and, one pipeline turn later, by its crop kernel; the buffer is written again two turns after that).
    h_img = torch.from_numpy(images).pin_memory()
    ring = [d_img] + [torch.empty_like(d_img) for _ in range(2)] if not args.resident else [d_img]
    copy_stream = torch.cuda.Stream()
    copied = {}

    def upload(k):
        """Enqueues the copy of batch number k of a run into ring slot k % 3, behind everything the compute stream
        holds so far (the last reader of that slot is among it)."""
        free = torch.cuda.Event()
        free.record()
        copy_stream.wait_event(free)
        with torch.cuda.stream(copy_stream):
            ring[k % 3].copy_(h_img, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        copied[k] = ev

    retried = []

    def run(batch_ids, queue=None):
        """Streams the batches named by `batch_ids` through the pipeline (detection of batch i+1 is enqueued while
        the host prepares the crops of batch i); every batch is complete, strings on the host, when it is yielded.
        With a `queue`, finished batches are reported to it and the batches in flight when a GPU call raises go back
        to it for another rank (this rank drains its stream and carries on with what the queue hands it next).
        -> [(batch id, {'counts', 'boxes', 'texts'})]"""
        from ocr_amd._lib import OcrHipError
        results = []
        ids = iter(batch_ids) if batch_ids is not None else None
        pending = []                # a batch taken while waiting for the queue to end: first of the next round

        def next_id():
            if pending:
                return pending.pop()
            # never wait for other ranks while batches are in flight here: they only complete when the feed moves on
            return next(ids, None) if queue is None else queue.take(block=False)
        while True:
            inflight = []

            def feed():
                k = 0
                if not args.resident:
                    copied.clear()
                    upload(0)
                while True:
                    i = next_id()
                    if i is None:
                        return
                    inflight.append(i)
                    if args.resident:
                        yield d_img, d_maps
                    else:
                        torch.cuda.current_stream().wait_event(copied.pop(k))
                        upload(k + 1)
                        yield ring[k % 3], d_maps
                    k += 1
            try:
                for out in pipe.run_batches(feed()):
                    i = inflight.pop(0)
                    results.append((i, {
                        'counts': [len(o['boxes']) for o in out],
                        'boxes': np.concatenate([o['boxes'] for o in out]) if out else np.zeros((0, 4), np.float32),
                        'texts': [t for o in out for t, v in zip(o['texts'], o['valid']) if v]}))
                    if queue is not None:
                        queue.done(i)
            except OcrHipError as e:
                if queue is None:
                    raise
                sys.stderr.write('bench.py: rank %d: %s; %d batch(es) go back to the queue\n' % (rank, e, len(inflight)))
                torch.cuda.synchronize()
                for i in inflight:
                    retried.append(i)
                    queue.give_back(i)
            if queue is None:
                return results
            # everything of this rank is done: now it may wait - for a batch another rank gives back, or for the end
            i = queue.take(block=True)
            if i is None:
                return results
            pending.append(i)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # Stage timers = two HIP events per launch, a few microseconds of stream time each (~90 launches per step: 1.5 % of
    # the step).  The warm-up runs with all of them and names the dominant matrix kernel; the timed region times only
    # that kernel (its launches, live, for `roofline`); the per-stage breakdown of the other kernels comes from an
    # untimed pass behind the timed region.
    def is_matrix(st):
        return st['flops'] > 0 and st['name'].startswith(('conv_mfma', 'conv_wino', 'conv_bf16', 'gemm_f32', 'gemm_bf16t'))

    pipe.ctx.profile_enable(True)
    run(range(max(1, args.warmup)))
    warm = pipe.ctx.profile_collect()
    dom_name = max((st for st in warm if is_matrix(st)), key=lambda st: st['total_ms'])['name']
    n_batches = world * args.steps
    queue = PageQueue(n_batches, grp if dist is not None else None)
    pipe.ctx.profile_only(dom_name)
    pipe.ctx.profile_enable(True)
    sync()
    t0 = time.perf_counter()
    local = run(None, queue)
    torch.cuda.synchronize()
    gathered = gather_results(local, n_batches, grp if dist is not None else None, dst=0)     # host gather, rank 0
    elapsed = time.perf_counter() - t0
    dom_live = [st for st in pipe.ctx.profile_collect() if st['name'] == dom_name]
    if dist is not None:
        dist.barrier()
    elapsed = max_over_ranks(elapsed, dist, None if args.share_gpu else device)
    my_steps = max(1, len(local))
    # the breakdown pass: every stage timed, a few steps, outside the clock
    breakdown_steps = min(4, args.steps)
    pipe.ctx.profile_only(None)
    pipe.ctx.profile_enable(True)
    run(range(breakdown_steps))
    stats = pipe.ctx.profile_collect()
    pipe.ctx.profile_enable(False)

    if rank == 0:
        pages = n_batches * B
        words = sum(len(r['texts']) for r in gathered)
        boxes = sum(sum(r['counts']) for r in gathered)
        acct = stage_accounting(pipe, d_img, d_maps)
        mfma = [s for s in stats if is_matrix(s)]
        dom = dom_live[0] if dom_live and dom_live[0]['launches'] else max(mfma, key=lambda s: s['total_ms'])
        direct = dom['flops'] / (dom['total_ms'] * 1e-3) / 1e12
        # a Winograd kernel is credited by the stage timers with the flop count of the convolution it computes
        # (2*9*Cin*Cout per output pixel); its matrix cores execute 36/144 (F(4x4)) or 16/36 (F(2x2)) of that, and only
        # the executed flops are bounded by the MFMA peak
        wino_frac = 0.25 if dom['name'].startswith('conv_wino4') else (16.0 / 36.0 if dom['name'].startswith('conv_wino') else 1.0)
        executed = direct * wino_frac
        peak = BF16_MFMA_PEAK_TFLOPS if dom['name'].startswith(('conv_bf16', 'gemm_bf16t')) else F32_MFMA_PEAK_TFLOPS
        gpu_ms = sum(s['total_ms'] for s in stats)
        conv_ms = sum(s['total_ms'] for s in mfma)
        conv_tflops = sum(s['flops'] for s in mfma) / (conv_ms * 1e-3) / 1e12
        line = {
            'metric': 'end-to-end OCR images/sec (%dx%d page)' % (S, S), 'value': pages / elapsed, 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.dtype == 'f32' else 'bf16 operands, f32 accumulate (non-default variant)', 'data': 'synthetic',
            'config': {'workload': 'configs[%d]: full EAST->crop->CRNN pipeline on synthetic %dx%d dense-text pages, %d-class CTC head'
                                   % (4 if args.vocab > 1000 else 3, S, S, args.vocab),
                       'pages_per_step_per_gpu': B, 'words_per_page': words / pages, 'boxes_per_page': boxes / pages,
                       'decode_candidates_per_page': acct['decode_candidates'] / B,
                       'detector_maps': 'teacher maps of the rendered pages feed decode; the random-weight EAST forward '
                                        'runs and is timed',
                       'parallelism': 'image-sharded x%d, one shared batch queue, boxes + strings gathered on the host of '
                                      'rank 0 inside the timed region' % world,
                       'bracket_collectives': ('none (one process)' if dist is None else
                                               {'nccl': 'rccl', 'gloo': 'gloo'}.get(dist.get_backend(), str(dist.get_backend()))),
                       'input': 'uint8 pages resident in HBM' if args.resident else
                                'pinned host uint8, H2D timed (copy stream, one batch ahead of the compute stream)',
                       'batches_taken_by_rank0': len(local), 'batches_given_back_by_rank0': len(retried)},
            **({'rehearsal': '--share-gpu: all ranks on one GPU, not a scaling result'} if args.share_gpu else {}),
            **({'variant': '--rotated: rotated-IoU NMS + rotated crops, an opt-in extension, not the reference path'}
               if args.rotated else {}),
            **({'variant': "--reference-defaults: BaseOCR.infer's default arguments (combine=True: one recogniser input per text "
                           "line; method='beam_search', num_beams=10 on the GPU); `words_per_s` counts paragraphs"}
               if args.reference_defaults else {}),
            'words_per_s': words / elapsed,
            'roofline': {'bound': 'mfma', 'kernel': dom['name'], 'launches': dom['launches'],
                         'avg_ms': dom['total_ms'] / dom['launches'], 'achieved': executed,
                         'peak': peak, 'unit': 'TFLOP/s', 'frac': executed / peak,
                         'direct_equiv_tflops': direct,
                         'algorithm': ('Winograd F(4x4,3x3): the matrix cores execute 36/144 of the direct convolution\'s flops; '
                                       '`achieved` counts the executed ones, `direct_equiv_tflops` the convolution computed')
                         if dom['name'].startswith('conv_wino4')
                         else ('Winograd F(2x2,3x3): the matrix cores execute 16/36 of the direct convolution\'s flops; '
                               '`achieved` counts the executed ones, `direct_equiv_tflops` the convolution computed')
                         if dom['name'].startswith('conv_wino') else 'direct implicit GEMM',
                         'traffic': pmc_traffic(dom['name']),
                         'algorithmic_bytes': dom['bytes'] / dom['launches'],
                         'executed_flop_per_launch': dom['flops'] * wino_frac / dom['launches']},
            'roofline_hbm': hbm_rooflines(stats, acct, args.vocab, breakdown_steps),
            'gpu_time': {'source': 'every launch timed in an untimed pass of %d steps behind the timed region (in the timed '
                                   'region only the dominant kernel carries events)' % breakdown_steps,
                         'kernels_ms_per_step': gpu_ms / breakdown_steps, 'all_mfma_conv_tflops_direct_equiv': conv_tflops,
                         'stages_ms_per_step': {s['name']: round(s['total_ms'] / breakdown_steps, 3) for s in stats},
                         'stages_tflops_direct_equiv': {s['name']: round(s['flops'] / (s['total_ms'] * 1e-3) / 1e12, 1)
                                                        for s in stats if s['flops'] > 0 and s['total_ms'] > 0}},
        }
        if world == 1 and not args.no_cpu_baseline and args.vocab == 97 and not args.rotated and not args.reference_defaults:
            line['cpu_baseline'] = cpu_baseline(images, maps, east_params, crnn_params, min(args.cpu_pages, B))
        print(json.dumps(line), flush=True)
    pipe.ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
