#!/bin/bash
# sustained bf16 MFMA ceiling of the box (scripts/probes/mfma_ceiling.hip) with rocm-smi clock / power samples next to it
# -> gpurun_out/mfma_ceiling.txt (copy to profiles/r06_mfma_ceiling.txt)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/probes/mfma_ceiling.hip -o scripts/probes/mfma_ceiling || exit 1
OUT=gpurun_out/mfma_ceiling.txt
SMI=gpurun_out/mfma_ceiling_smi.txt
: > $SMI
( while true; do echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr '\n' ' ')" >> $SMI; sleep 0.5; done ) &
SAMPLER=$!
sleep 1.5
echo "t_start=$(date +%s.%N | cut -c1-14)" > $OUT
scripts/probes/mfma_ceiling ${1:-10} >> $OUT
echo "t_end=$(date +%s.%N | cut -c1-14)" >> $OUT
sleep 1
kill $SAMPLER
echo "# rocm-smi samples (0.5 s apart; idle before / after the probe included)" >> $OUT
sed -e 's/  */ /g' $SMI >> $OUT
cat $OUT
