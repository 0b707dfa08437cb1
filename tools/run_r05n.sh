mkdir -p gpurun_out
{
echo "# Where a workgroup's time goes (wave 0 of every workgroup, -DISING_FUSED_TRACE build): the fused form at the library's shape against split launches at theirs (ISING_SPLIT=1, the"
echo "# strip height ising_create picks for long calls), same box.  RESULT = flips/ns of the trace build, strip rows, layout, counts + bond sum after 96 sweeps (equal across forms)."
for cfg in "16384 16384 16" "8192 8192 4" "65536 8192 16"; do
  set -- $cfg
  echo "== $2 x $1, the fused form at the library's shape (ISING_SPLIT=0)"
  ISING_LIB=$PWD/ising_gpu_amd/libising_hip_trace.so ISING_SPLIT=0 python tools/ab_probe.py case $1 $2 0 2>&1 | grep -v amdgpu.ids
  echo "== $2 x $1, split launches, strips of $3 rows, five workgroups per CU"
  ISING_LIB=$PWD/ising_gpu_amd/libising_hip_trace.so ISING_SPLIT=1 python tools/ab_probe.py case $1 $2 $3 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/split_vs_fused_trace_r05.txt 2>&1
