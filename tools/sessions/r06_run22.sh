#!/bin/bash
# round 6, call 26: C5, the later rows of a block in ONE launch (OGSQP_WIDE_INBLOCK=1) against the three launches, 30 iterations each
for v in 0 1 0 1; do
  OGSQP_WIDE_INBLOCK=$v python tools/sqp_solve.py launch4 30 1e-6 hip 2>&1 | tail -1 | sed "s/^/inblock=$v /" | cut -c1-330
done
