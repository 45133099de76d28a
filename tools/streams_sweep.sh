#!/bin/bash
# usage: streams_sweep.sh lib scene spp "K1 K2 ..." [width height]   (PT_STREAMS sweep of one library, best of two frames each)
lib=$1; scene=$2; spp=$3; w=${5:-1024}; h=${6:-768}
for k in $4; do
PT_STREAMS=$k PT_LIB=$PWD/$lib timeout -k 10 120 python3 -c "
import sys; sys.path.insert(0, 'tools'); import ab_flags as f
f.W, f.H = $w, $h
img, n, t = f.render('$scene', $spp, 0, 0, reps=2)
print('$lib ${w}x$h K=$k m=%d %.1f ms %.3f G bounces/s' % (-(-$w*$h//$k), t*1e3, n/t/1e9))
" || exit 1
done
