#!/bin/bash
# here (not on the GPU box): gpurun_out/final_$TAG + prof_$TAG + the probes' outputs -> profiles/ (what DESIGN.md cites)
#   tools/publish_profiles.sh r03b        (files are named r03_* / r03b_*: the round is the tag without its letter)
TAG=${1:-r03a}
R=${TAG%[a-z]}
F=gpurun_out/final_$TAG
for f in batch_scaling batch_scaling_small e2e_chunks pcie_probe enc_scaling; do
  [ -f $F/$f.txt ] && grep -v amdgpu.ids $F/$f.txt > profiles/${R}_$f.txt
done
for w in C2 C3 C4 C5 C5_strong_1gpu C4_2rank_rehearsal C3_strong_2rank_rehearsal; do [ -f $F/bench_$w.json ] && grep '^{' $F/bench_$w.json > profiles/${R}_final_bench_$w.json; done   # (gloo prints its connection lines to stdout)
[ -f gpurun_out/sq_mix/mix.txt ] && cp gpurun_out/sq_mix/mix.txt profiles/${R}_sq_instruction_mix.txt
[ -f gpurun_out/pmc_records/records.txt ] && cat gpurun_out/pmc_records/records.txt gpurun_out/pmc_records/probe.log | grep -v amdgpu.ids > profiles/${R}_residual_pmc_records_pass.txt
[ -f gpurun_out/${R}_exp_residual.txt ] && cp gpurun_out/${R}_exp_residual.txt profiles/${R}_exp_residual.txt
python3 tools/pmc_summary.py $TAG > /dev/null
# one tag of rocprof summaries per round is kept
for old in profiles/${R}?_*; do case "$old" in profiles/${TAG}_*) ;; *) [ -e "$old" ] && { git rm -q --cached "$old" 2>/dev/null; rm -f "$old"; };; esac; done
for old in profiles/pmc_traffic_${R}?.json; do [ -e "$old" ] && [ "$old" != "profiles/pmc_traffic_$TAG.json" ] && { git rm -q --cached "$old" 2>/dev/null; rm -f "$old"; }; done
ls profiles | grep -c .
