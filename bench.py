"""End-to-end OCR throughput on synthetic 1024x1024 dense-text pages (BASELINE.json config 4).

    python bench.py --gpus 1 --steps 40 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over a batch of pages already resident in HBM:
EAST-VGG16 forward (f32 MFMA) -> RBOX decode -> LANMS + NMS -> crop/resize/pack -> CRNN + CTC greedy
-> strings on the host.  Images are independent, so ranks share nothing (weak scaling, no
collective in the data path; one barrier + one MAX all-reduce bracket the timed region).

No trained weights exist offline: both networks run with seeded random weights of the reference
architectures (all their arithmetic is executed and timed), and the decode stage consumes the
synthetic "teacher" maps of the rendered pages instead of the random-weight network output, so that
decode / LANMS / crop / CRNN see realistic box counts (~18 000 candidates and ~200 words per page).

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (the 3x3 f32-MFMA
convolution), timed with HIP events on the launch stream inside the timed region by the library's
stage timers; `cpu_baseline` times the reference-faithful CPU path (oracle/) on the host cores.
"""
import argparse
import json
import os
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
    ap.add_argument('--batch', type=int, default=16, help='pages per step and per GPU')
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--words', type=int, default=220)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="f32 = the reference's precision (the headline); bf16 = bfloat16 operands, float32 accumulation")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-pages', type=int, default=4)
    return ap.parse_args()


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
    return {'value': n_pages / dt, 'unit': 'images/s', 'cores': int(torch.get_num_threads()), 'kind': 'port',
            'words_per_s': words / dt,
            'sample': '%d page(s) %dx%d of the GPU workload through oracle/pipeline.py: batch-1 EAST (torch-CPU f32), '
                      'NumPy decode, C LANMS (1 thread), one CRNN forward per text row; %.1f s'
                      % (n_pages, images.shape[1], images.shape[2], dt)}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed counter passes of this same command
    (tools/pmc_traffic.py over `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`), or None."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fh:
            return json.load(fh)['kernels'][kernel]['traffic_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def main():
    args = parse()
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    from ocr_amd.dist import init_from_env, max_over_ranks
    rank, world, dist = init_from_env('nccl', torch.device('cuda', local_rank))     # RCCL when launched by torchrun

    from ocr_amd import synth
    from ocr_amd.pipeline import OcrPipeline
    from ocr_amd.weights import random_east_params, random_crnn_params

    east_params = random_east_params(seed=1)
    crnn_params = random_crnn_params(vocab_size=97, seed=2)
    from ocr_amd._lib import BF16, F32
    pipe = OcrPipeline(east_params, crnn_params, device=local_rank, compute_dtype=BF16 if args.dtype == 'bf16' else F32)

    B, S = args.batch, args.size
    images, maps, _ = synth.make_pages(1234 + 4 + 100 * rank, B, S, S, args.words)
    d_img = torch.from_numpy(images).cuda()
    d_maps = torch.from_numpy(maps).cuda()

    def run(n_steps):
        # n_steps batches streamed through the pipeline (detection of batch i+1 is enqueued while the host
        # prepares the crops of batch i); every batch is completed, strings included, before run() returns
        w = bx = 0
        for out in pipe.run_batches((d_img, d_maps) for _ in range(n_steps)):
            w, bx = sum(int(o['valid'].sum()) for o in out), sum(len(o['boxes']) for o in out)
        return w, bx

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    words = boxes = 0
    if args.warmup:
        words, boxes = run(args.warmup)
    pipe.ctx.profile_enable(True)
    sync()
    t0 = time.perf_counter()
    words, boxes = run(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    stats = pipe.ctx.profile_collect()
    pipe.ctx.profile_enable(False)
    if dist is not None:
        dist.barrier()
    elapsed = max_over_ranks(elapsed, dist, torch.device('cuda', local_rank))

    if rank == 0:
        pages = world * args.steps * B
        mfma = [s for s in stats if s['flops'] > 0 and s['name'].startswith(('conv_mfma', 'conv_wino', 'conv_bf16', 'gemm_f32'))]
        dom = max(mfma, key=lambda s: s['total_ms'])
        achieved = dom['flops'] / (dom['total_ms'] * 1e-3) / 1e12
        # the Winograd kernel is credited with the flop count of the convolution it computes (2*9*Cin*Cout per output
        # pixel) but its matrix cores execute 16/36 of that: `executed` is what the MFMA peak actually bounds
        wino_frac = 0.25 if dom['name'].startswith('conv_wino4') else (16.0 / 36.0 if dom['name'].startswith('conv_wino') else 1.0)
        executed = achieved * wino_frac
        peak = BF16_MFMA_PEAK_TFLOPS if dom['name'].startswith('conv_bf16') else F32_MFMA_PEAK_TFLOPS
        gpu_ms = sum(s['total_ms'] for s in stats)
        conv_ms = sum(s['total_ms'] for s in mfma)
        conv_tflops = sum(s['flops'] for s in mfma) / (conv_ms * 1e-3) / 1e12
        line = {
            'metric': 'end-to-end OCR images/sec (1024x1024 page)', 'value': pages / elapsed, 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.dtype == 'f32' else 'bf16 operands, f32 accumulate (non-default variant)', 'data': 'synthetic',
            'config': {'workload': 'configs[3]: full EAST->crop->CRNN pipeline on synthetic %dx%d dense-text pages' % (S, S),
                       'pages_per_step_per_gpu': B, 'words_per_page': words / B, 'decode_candidates_per_page': None,
                       'detector_maps': 'teacher maps of the rendered pages feed decode; the random-weight EAST forward '
                                        'runs and is timed', 'parallelism': 'image-sharded x%d' % world},
            'words_per_s': world * args.steps * words / elapsed,
            'roofline': {'bound': 'mfma', 'kernel': dom['name'], 'launches': dom['launches'],
                         'avg_ms': dom['total_ms'] / dom['launches'], 'achieved': achieved,
                         'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                         'executed_tflops': executed, 'executed_frac': executed / peak,
                         'algorithm': ('Winograd F(4x4,3x3): `achieved` counts the direct convolution\'s flops, the matrix '
                                       'cores execute 36/144 of them (`executed`)') if dom['name'].startswith('conv_wino4')
                         else ('Winograd F(2x2,3x3): `achieved` counts the direct convolution\'s flops, the matrix '
                               'cores execute 16/36 of them (`executed`)') if dom['name'].startswith('conv_wino')
                         else 'direct implicit GEMM',
                         'traffic': pmc_traffic(dom['name']),
                         'algorithmic_bytes': dom['bytes'] / dom['launches'],
                         'flop_per_launch': dom['flops'] / dom['launches']},
            'gpu_time': {'kernels_ms_per_step': gpu_ms / args.steps, 'all_mfma_conv_tflops': conv_tflops,
                         'stages_ms_per_step': {s['name']: round(s['total_ms'] / args.steps, 3) for s in stats}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(images, maps, east_params, crnn_params, min(args.cpu_pages, B))
        print(json.dumps(line), flush=True)
    pipe.ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
