#!/usr/bin/env python3
"""Markdown table of tools/bench_conv_f32.py's JSON (own fp32 convolution kernels against MIOpen per ResNet-50 layer shape), with the sums over the
distinct shapes and weighted by how often each shape occurs in one ResNet-50 step.  usage: python tools/conv_layers_md.py layers.json [previous.json] > profiles/rNN_conv_f32_layers.md"""
import json, sys
OCC = {'stem': 1, 'l1.c1a': 1, 'l1.c2': 3, 'l1.c3': 4, 'l1.c1': 2, 'l2.c1a': 1, 'l2.c2s': 1, 'l2.c3': 4, 'l2.ds': 1, 'l2.c1': 3, 'l2.c2': 3, 'l3.c1a': 1, 'l3.c2s': 1, 'l3.c3': 6,
       'l3.ds': 1, 'l3.c1': 5, 'l3.c2': 5, 'l4.c1a': 1, 'l4.c2s': 1, 'l4.c3': 3, 'l4.ds': 1, 'l4.c1': 2, 'l4.c2': 2}
rows = json.load(open(sys.argv[1]))
prev = {r['layer']: r for r in json.load(open(sys.argv[2]))} if len(sys.argv) > 2 else {}
print('| layer | Cin -> Cout, k, stride @ HxW | x n | GFLOP | fwd own / lib | dgrad own / lib | wgrad own / lib%s | fwd TFLOP/s |' % (' (round 3 own)' if prev else ''))
print('|---|---|---|---|---|---|---|---|')
S = {k: 0.0 for k in ('fo', 'fl', 'do', 'dl', 'wo', 'wl')}; Wt = dict(S)
lose = []
for r in rows:
    n = OCC.get(r['layer'], 1)
    f = lambda a, b: '%.0f / %.0f' % (r[a], r[b]) if a in r and b in r else '- / -'
    pw = ' (%.0f)' % prev[r['layer']]['wgrad'] if r['layer'] in prev else ''
    print('| %s | %d -> %d, %dx%d, /%d @ %d | %d | %.1f | %s | %s | %s%s | %.1f |' % (r['layer'], r['cin'], r['cout'], r['k'], r['k'], r['stride'], r['hw'], n, r['gflop'],
          f('fwd_stats', 'lib_fwd'), f('dgrad', 'lib_dgrad'), f('wgrad', 'lib_wgrad'), pw, r['tflops_fwd']))
    for key, a, b in (('f', 'fwd_stats', 'lib_fwd'), ('d', 'dgrad', 'lib_dgrad'), ('w', 'wgrad', 'lib_wgrad')):
        if a in r and b in r:
            S[key + 'o'] += r[a]; S[key + 'l'] += r[b]; Wt[key + 'o'] += n * r[a]; Wt[key + 'l'] += n * r[b]
            if r[a] > 1.02 * r[b]:
                lose.append('%s %s %.0f / %.0f' % (r['layer'], {'f': 'fwd', 'd': 'dgrad', 'w': 'wgrad'}[key], r[a], r[b]))
ms = lambda v: v / 1e3
print('\nSum over the distinct shapes: forward %.1f ms own / %.1f library, data gradient %.1f / %.1f, weight gradient %.1f / %.1f -- **%.1f ms own against %.1f library**.'
      % (ms(S['fo']), ms(S['fl']), ms(S['do']), ms(S['dl']), ms(S['wo']), ms(S['wl']), ms(S['fo'] + S['do'] + S['wo']), ms(S['fl'] + S['dl'] + S['wl'])))
print('Weighted by occurrence (one ResNet-50 step of 512 rows on one stream): forward %.1f / %.1f ms, data gradient %.1f / %.1f ms, weight gradient %.1f / %.1f ms = %.1f ms own against %.1f.'
      % (ms(Wt['fo']), ms(Wt['fl']), ms(Wt['do']), ms(Wt['dl']), ms(Wt['wo']), ms(Wt['wl']), ms(Wt['fo'] + Wt['do'] + Wt['wo']), ms(Wt['fl'] + Wt['dl'] + Wt['wl'])))
print('(`fwd own` is the forward WITH the BatchNorm statistics in its epilogue, which the library kernel does not compute.)')
print('\nShapes where the own kernel is more than 2 %% slower than the library (%d): %s' % (len(lose), '; '.join(lose) if lose else 'none'))
