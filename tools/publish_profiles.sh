#!/bin/bash
# here (not on the GPU box): gpurun_out/final_$TAG + prof_$TAG -> profiles/ (what DESIGN.md cites)
TAG=${1:-r02c}
F=gpurun_out/final_$TAG
for f in batch_scaling e2e_chunks pcie_probe enc_scaling ubench_mix ubench_lat ubench_v6_waves ubench_ctx ubench_ops; do
  [ -f $F/$f.txt ] && grep -v amdgpu.ids $F/$f.txt > profiles/r02_$f.txt
done
for w in C2 C3 C4 C5 C5_strong_1gpu C4_2rank_rehearsal; do cp $F/bench_$w.json profiles/r02_final_bench_$w.json; done
[ -f gpurun_out/parse_profile.txt ] && grep -v amdgpu.ids gpurun_out/parse_profile.txt > profiles/r02_parse_profile.txt
[ -f gpurun_out/sq_mix/mix.txt ] && cp gpurun_out/sq_mix/mix.txt profiles/r02_sq_instruction_mix.txt
[ -f gpurun_out/pmc_tcc/tcc.txt ] && cp gpurun_out/pmc_tcc/tcc.txt profiles/r02_residual_pmc_tcc_requests.txt
python3 tools/pmc_summary.py $TAG > /dev/null
# one tag of rocprof summaries is kept
for old in profiles/r02?_*; do case "$old" in profiles/${TAG}_*) ;; *) git rm -q --cached "$old" 2>/dev/null; rm -f "$old";; esac; done
for old in profiles/pmc_traffic_r02?.json; do [ "$old" != "profiles/pmc_traffic_$TAG.json" ] && { git rm -q --cached "$old" 2>/dev/null; rm -f "$old"; }; done
ls profiles | grep -c .
