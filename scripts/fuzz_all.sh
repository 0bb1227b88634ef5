# the wide parity sweeps whose totals DESIGN.md quotes; writes gpurun_out/fuzz_summary.txt (copied to profiles/rNN_fuzz_summary.txt)
out=gpurun_out/fuzz_summary.txt
echo "wide parity sweeps, GPU (libboofhip.so through the C ABI) against the CPU oracle; $(date -u +%Y-%m-%dT%H:%MZ); commit ${FUZZ_COMMIT:-unknown}" > $out
S1=${FUZZ_SEED1:-301}; S2=${FUZZ_SEED2:-302}
for job in "fuzz_parity $S1 ${FUZZ_PARITY_CASES:-1200}" "fuzz_parity $S2 ${FUZZ_PARITY_CASES:-1200}" "fuzz_assoc $S1 ${FUZZ_ASSOC_CASES:-3000}" "fuzz_assoc $S2 ${FUZZ_ASSOC_CASES:-3000}" "fuzz_ip $S1 ${FUZZ_IP_CASES:-1500}" "fuzz_ip $S2 ${FUZZ_IP_CASES:-1500}"; do
  set -- $job
  echo "== python scripts/$1.py $2 $3" >> $out
  timeout -k 10 1000 python scripts/$1.py $2 $3 > gpurun_out/$1_$2.log 2>&1
  echo "   rc=$? ; $(grep -c MISMATCH gpurun_out/$1_$2.log) MISMATCH lines, $(grep -c EXCEPTION gpurun_out/$1_$2.log) EXCEPTION lines ; $(tail -1 gpurun_out/$1_$2.log)" >> $out
done
cat $out
