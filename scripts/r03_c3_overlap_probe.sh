#!/bin/bash
# how much of G-Beams' build + traversal could hide beside its evaluation: one process against two sharing the GPU
one() { python bench.py --workload c3 --only-timed --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 %.0f Mevals/s %.2f ms/step' % (d['value'], d['ms_per_step']))"; }
one alone
one "pair a" & one "pair b" & wait
