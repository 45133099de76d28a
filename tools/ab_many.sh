#!/bin/bash
# usage: ab_many.sh "lib1 lib2 ..." scene spp rounds
libs="$1"; scene=${2:-cornell}; spp=${3:-2048}; rounds=${4:-2}
for r in $(seq $rounds); do for l in $libs; do
PT_LIB=$PWD/$l timeout -k 10 120 python3 -c "
import sys; sys.path.insert(0, 'tools'); import ab_flags as f; import numpy as np
img, n, t = f.render('$scene', $spp, 0, int('${AB_BACKEND:-0}'), reps=2)
print('%-40s %s %.1f ms %.3f G bounces/s hash %016x' % ('$l', '$scene', t*1e3, n/t/1e9, int(np.bitwise_xor.reduce(img.view(np.uint32).astype(np.uint64) * np.arange(1, img.size+1, dtype=np.uint64)))))
" || exit 1
done; done
