set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2s2_pp_timeline.txt
: > $O
for spec in "150528 2304 768 plain" "150528 2304 768 plain pp_cont=0" "150528 768 768 scale" "150528 768 768 residual" "150528 768 768 residual pp_cont=0" "150624 3072 768 mul" "150624 3072 768 mul pp_cont=0" "150624 3072 768 gelu2" "150624 768 3072 plain"; do
  timeout 120 python tools/pp_timeline.py $spec >> $O 2>&1
  echo >> $O
done
grep -v amdgpu.ids $O
