#!/bin/bash
# ab.sh <out dir> "<task [bench args]>" <lib> [<lib> ...] — bench lines of library variants back to back on one box, two rounds
O=$1; T="$2"; shift; shift; mkdir -p $O
C=loco_mujoco_amd/csrc
for R in 1 2; do for V in "$@"; do
  LOCOHIP_LIB=$PWD/$C/$V timeout 600 python bench.py --task $T --steps 100 --warmup 30 --no-cpu-baseline --configs off --surface-steps 0 2>$O/err_$V.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$T $V: %.3f ms/step %.0f env-steps/s' % (d['ms_per_step'], d['value']))" >> $O/ab.log 2>&1
done; done
cat $O/ab.log
