mkdir -p gpurun_out
bash tools/run_baseline_configs.sh > gpurun_out/configs_r05.txt 2>&1
CLI=ising_gpu_amd/cuIsing
{
echo "### mid-size lattices through the CLI, T=Tc, seed 1234 (split launches where ising_create picks them; ISING_SPLIT=0: the fused form)"
for shape in "16384 16384 20000" "8192 8192 40000" "65536 8192 8192" "8192 16384 20000" "24576 24576 4096" "32768 4096 20000"; do
  set -- $shape
  for p in 0 16; do
    for env in "" "ISING_SPLIT=0"; do
      r=$(env $env $CLI -x $1 -y $2 -d 1 -n $3 -a 1 -s 1234 $( [ $p = 16 ] && echo "-p 16" ) | grep -E "Kernel execution|Final" | tr '\n' ' ')
      echo "$2 x $1, $3 sweeps, ${env:-default}$( [ $p = 16 ] && echo ", -p 16" ): $r"
    done
  done
done
} > gpurun_out/midsize_cli_r05.txt 2>&1
