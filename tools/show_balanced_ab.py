#!/usr/bin/env python3
"""Print tools/bench_conv_f32_balanced.py's json as a table.  usage: python tools/show_balanced_ab.py ab.json"""
import json, sys
d = json.load(open(sys.argv[1]))
for r in d['layers']:
    for k in ('fwd', 'dgrad'):
        a, b, c = r[k + '_us_0'], r[k + '_us_2'], r[k + '_us_1']
        if a != a:
            continue
        t = r['tiles_' + k]
        print('%-7s %-5s tiles %5d rounds %5.2f  tile walk %6.1f us  balanced %6.1f (%+5.1f %%)  launcher %6.1f (%+5.1f %%)'
              % (r['layer'], k, t, t / 512, a, b, 100 * (b / a - 1), c, 100 * (c / a - 1)))
print(d['weighted_total_ms'])
