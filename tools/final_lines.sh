set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python bench.py > gpurun_out/r04_final_cornell4096.json 2> gpurun_out/r04_final_cornell4096.err
python bench.py --spp 1024 --no-cpu-baseline --no-variants > gpurun_out/r04_final_cornell1024.json 2>/dev/null
python bench.py --scene mesh --spp 1024 --no-cpu-baseline --no-variants > gpurun_out/r04_final_mesh1024.json 2>/dev/null
for cfg in "cornell 450 300 500" "cornell 900 600 1000" "cornell 3000 2000 100" "cornell 1024 768 4096" "mesh 450 300 500" "mesh 900 600 1000" "mesh 3000 2000 100" "mesh 1024 768 1024"; do set -- $cfg; python bench.py --no-variants --no-cpu-baseline --scene $1 --width $2 --height $3 --spp $4 --steps 10 --warmup 5 > gpurun_out/r04_size_$1_$2x$3_$4.json 2>/dev/null; done
for l in 256:cornell 128:mesh; do PT_LIB=path-tracer-rust_amd/libptrace_hip.so tools/env_sweep2.sh path-tracer-rust_amd/libptrace_hip.so ${l#*:} ${l%%:*} PT_RAYS_PER_PASS 536870912; done > gpurun_out/r04_shipped_rates.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_r04a2 -- python3 $R/bench.py --no-cpu-baseline --no-variants --rays-per-pass 536870912 > $R/gpurun_out/stats_r04a2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_r04m2 -- python3 $R/bench.py --no-cpu-baseline --no-variants --scene mesh --spp 1024 --rays-per-pass 536870912 > $R/gpurun_out/stats_r04m2.log 2>&1
cd $R
python bench.py --width 4096 --height 4096 --spp 16384 --steps 1 --warmup 0 --no-cpu-baseline --no-variants > gpurun_out/r04_final_config5_4096x4096_16384spp_1gpu.json 2>/dev/null
echo lines done
