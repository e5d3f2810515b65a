R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s; mkdir -p $O; cd $R
python scripts/dev/stress_bias_lds.py 200 2>&1 | grep -v amdgpu.ids | tee $O/bias_lds_product.txt
for k in "$@"; do
  MAGNET_LIB_PATH=$R/magnet_amd/csrc/variants/libmagnet_hip_bl$k.so timeout 300 python scripts/dev/stress_bias_lds.py 200 2>&1 | grep -v amdgpu.ids | tee $O/bias_lds_bl$k.txt
done
