#!/bin/bash
# compile one csrc file with resource-usage remarks (container: no GPU needed); usage: cc_nt.sh gemm_nt [kernel-name-filter]
f=${1:-gemm_nt}; filt=${2:-pp_kernel}
cd /root/repo/videotransformer-pytorch_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-unused-value -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/$f.o 2>&1 \
 | grep -E "error|Function Name|VGPRs:|Spill|ScratchSize|SGPRs:" | paste - - - - - - 2>/dev/null | grep -E "error|$filt" \
 | sed 's/\[-Rpass[^]]*\]//g; s/[a-z_]*.hip:[0-9]*:[0-9]*: remark://g; s/EEEviii.*EpiParamsE//' | sed 's/Function Name: //' | cut -c1-220
