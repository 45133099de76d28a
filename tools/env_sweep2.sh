#!/bin/bash
# usage: env_sweep2.sh lib scene spp VAR "v1 v2 ..." [width height]   (sweep of one tuning variable of one library, best of two frames each)
lib=$1; scene=$2; spp=$3; var=$4; w=${6:-1024}; h=${7:-768}
for v in $5; do
env $var=$v PT_LIB=$PWD/$lib timeout -k 10 120 python3 -c "
import sys; sys.path.insert(0, 'tools'); import ab_flags as f
f.W, f.H = $w, $h
img, n, t = f.render('$scene', $spp, 0, 0, reps=2)
print('$lib ${w}x$h $scene@$spp $var=$v %.1f ms %.3f G bounces/s' % (t*1e3, n/t/1e9))
" || exit 1
done
