#!/usr/bin/env python3
"""profiles/r01_bn_pmc.{json,md} from the four summaries `tools/summarize_pmc.py` writes (FETCH_SIZE / WRITE_SIZE passes of
`bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-stress`, prefixes lec::bn_ and lec::conv).

    python tools/make_bn_pmc_profile.py gpurun_out <algorithmic GB per step from bench.py's roofline.alg_bytes_per_step>"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    d, alg = sys.argv[1], float(sys.argv[2])
    L = lambda f: json.load(open(os.path.join(d, f)))['bytes_per_step']
    bf, bw, cf, cw = L('pmc_bn_FETCH_SIZE.json'), L('pmc_bn_WRITE_SIZE.json'), L('pmc_conv_FETCH_SIZE.json'), L('pmc_conv_WRITE_SIZE.json')
    tot = 2 * sum(bf.values()) + sum(bw.values()); ctot = 2 * sum(cf.values()) + sum(cw.values())
    json.dump({'traffic_bytes_per_step_fetch_x2': tot, 'fetch_raw': bf, 'write': bw,
               'mfma_convolutions': {'fetch_raw': cf, 'write': cw, 'traffic_bytes_per_step_fetch_x2': ctot}},
              open(os.path.join(ROOT, 'profiles', 'r01_bn_pmc.json'), 'w'), indent=1)
    g = lambda t, k: t.get(k, 0.0) / 1e9
    row = lambda t1, t2, k, note: '| `%s` | %.2f | %.2f | %.2f | %s |' % (k, g(t1, k), 2 * g(t1, k), g(t2, k), note)
    md = ['# Fused BatchNorm family (and the MFMA convolutions): HBM traffic per bench step from rocprofv3 PMC (round 1, MI355X, final build)', '',
          '`rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-stress` and a '
          'second pass with `--pmc WRITE_SIZE`, summed per kernel and divided by the steps of the run with `tools/summarize_pmc.py`, assembled by '
          '`tools/make_bn_pmc_profile.py` (config 3, 512 CNN rows, 5.43 G elements over 53 BatchNorm layers; 16 block outputs carry two gradient '
          'streams).  FETCH_SIZE on gfx950 reports half of the bytes of wide (16 B/lane) streaming reads (MI355X guide, HBM section): the x2 column '
          'is the corrected figure; WRITE_SIZE is exact for 16-byte streaming stores.', '',
          '| kernel | FETCH_SIZE raw GB | read GB (x2) | WRITE_SIZE GB | what it moves |', '|---|---|---|---|---|',
          row(bf, bw, 'lec::bn_stats_kernel', 'x of the 35 layers whose producer leaves no statistics'),
          row(bf, bw, 'lec::bn_apply_kernel', 'x + residual in, y + mask out (46 layers: the other 7 are applied in the epilogue of their convolution)'),
          row(bf, bw, 'lec::bn_bwd_reduce_kernel', 'pass 1 of the 47 layers that still run it: dy + second gradient stream + x + mask in, g (= d residual) of the residual layers out'),
          row(bf, bw, 'lec::bn_bwd_apply_kernel', 'dy or g + x (+ mask) in, dx out (46 layers: the other 7 run inside their convolution\'s weight gradient)'),
          row(bf, bw, 'lec::bn_stats_finalize_kernel', 'partials'), row(bf, bw, 'lec::bn_bwd_finalize_kernel', 'partials'), '',
          'Total corrected BatchNorm traffic **%.1f GB per step** against %.1f GB algorithmic (`roofline.alg_bytes_per_step` of `bench.py`): no wasted '
          're-reads.  History: 115.6 GB with both backward passes re-reading dy, the second gradient stream and the mask; 110.1 GB once pass 1 wrote '
          'the masked gradient of the residual layers; 103.2 GB once 18 statistics passes were gone (their sums come out of the producing '
          'convolution\'s epilogue); 91.5 GB with pass 1 of five forked block outputs in the epilogue of the next conv1 data gradient '
          '(`lec_conv1x1_dgrad_bnfold`); 88.2 GB with the layer1 -> layer2 transition folded as well; %.1f GB with the apply pass of the seven conv3 -> bn3 '
          'layers moved into a second run of the convolution (`lec_conv1x1_fwd_bnapply`) and pass 2 of their backward into the convolution\'s weight-gradient '
          'kernel (`lec_conv1x1_wgrad_bnapply`, counted with the convolutions below).  The convolution kernels read the second gradient, x and the '
          'mask of the six folded tensors and the residual of the seven applied ones instead: their FETCH_SIZE went from 4.63 to %.2f GB raw.' % (tot / 1e9, alg, tot / 1e9, g(cf, 'lec::conv1x1_fwd_stats_kernel')), '',
          'MFMA convolutions (`conv_mfma.hip`: the 1x1 kernels, forward with statistics and data gradient of the wide layers, and layer1\'s 3x3 halo kernel):', '',
          '| kernel | FETCH_SIZE raw GB | read GB (x2) | WRITE_SIZE GB | |', '|---|---|---|---|---|',
          row(cf, cw, 'lec::conv1x1_bigk_kernel', ''), row(cf, cw, 'lec::conv1x1_fwd_stats_kernel', ''), row(cf, cw, 'lec::conv3x3_c64_halo_kernel', ''), '',
          'and, summed under the prefix `lec::wgrad` by a separate `summarize_pmc.py` call when wanted, the weight-gradient kernel of the seven conv3 layers (g, x of the BatchNorm and the layer input in; dx out).', '',
          'The 3x3 halo kernel reads 0.21 GB algorithmic per launch (6 launches); its halo (60 input pixels per 32 outputs) is served by the XCD\'s L2 for the most part.', '']
    open(os.path.join(ROOT, 'profiles', 'r01_bn_pmc.md'), 'w').write('\n'.join(md))
    print('BN traffic %.2f GB/step (algorithmic %.2f), MFMA convolutions %.2f GB/step' % (tot / 1e9, alg, ctot / 1e9))


if __name__ == '__main__':
    main()
