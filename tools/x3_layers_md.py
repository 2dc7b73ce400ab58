#!/usr/bin/env python3
"""Per-layer table of the fp32 convolution kernels: split (conv_f32x3.hip) / f32-MFMA (conv_f32.hip) / MIOpen, microseconds and error vs float64.
usage: python tools/x3_layers_md.py > profiles/rNN_conv_f32x3_layers.md   (runs tools/check_x3.py --wgrad on the 23 ResNet-50 shapes)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_x3.py'), '--wgrad', '--all', '--lib-time'], capture_output=True, text=True).stdout
rows = [json.loads(l) for l in out.splitlines() if l.startswith('{')]
print('# fp32 convolutions per layer shape: split kernels (csrc/conv_f32x3.hip) against the f32-MFMA kernels (csrc/conv_f32.hip) and MIOpen fp32\n')
print('`python tools/x3_layers_md.py` on one MI355X: the 23 distinct convolution shapes of ResNet-50 at 224 x 224, 512 images; microseconds per launch (HIP events,')
print('3 back-to-back launches) as split / f32-MFMA / MIOpen; error = max |result - float64| / max |float64| on 64 images as split / f32-MFMA / MIOpen.')
print('`-`: the direction does not exist (stem data gradient) or the split kernel does not serve the shape (weight gradient below 128 channels: the f32-MFMA kernel runs).\n')
print('| layer | fwd us | dgrad us | wgrad us | fwd err | dgrad err | wgrad err |\n|---|---|---|---|---|---|---|')
tot = {k: 0.0 for k in ('fwd_x3', 'fwd_f32', 'fwd_lib', 'dgrad_x3', 'dgrad_f32', 'dgrad_lib', 'wgrad_x3', 'wgrad_f32', 'wgrad_lib')}
for r in rows:
    def cell(d, kind='us'):
        ks = ['%s_%s_%s' % (d, kind, w) for w in ('x3', 'f32', 'lib')]
        if ks[0] not in r and ks[1] not in r:
            return '-'
        return ' / '.join(str(r.get(k, '-')) for k in ks)
    n = r.get('count', 1)
    for d in ('fwd', 'dgrad', 'wgrad'):
        for w in ('x3', 'f32', 'lib'):
            v = r.get('%s_us_%s' % (d, w), r.get('%s_us_f32' % d, 0.0) if w == 'x3' else 0.0)
            tot['%s_%s' % (d, w)] += n * (v or 0.0)
    print('| %s (x%d) | %s | %s | %s | %s | %s | %s |' % (r['layer'], n, cell('fwd'), cell('dgrad'), cell('wgrad'), cell('fwd', 'err'), cell('dgrad', 'err'), cell('wgrad', 'err')))
print('\nWeighted by occurrence (one ResNet-50 step, ms; where the split kernel does not serve a shape the f32-MFMA time counts): ' +
      '; '.join('%s %.1f / %.1f / %.1f' % (d, tot[d + '_x3'] / 1e3, tot[d + '_f32'] / 1e3, tot[d + '_lib'] / 1e3) for d in ('fwd', 'dgrad', 'wgrad')) + '.')
