cd $GRAFT_REPO_ROOT
for d in "5" "1 3" "0"; do echo "== dropped $d"; timeout 200 python tools/micro/ffn_compact_debug.py $d 2>&1 | grep -v amdgpu.ids; done
