# Runs on the GPU box: hashes of what an iteration / a ChamferDistance call produce (scripts/exp/step_bits.py) with the tree's library,
# with the vector-unit approximate pass, and -- REF_LIB=build/variants/libsaga_X.so -- with another build; compared with each other and
# with scripts/exp/step_bits_reference.txt (the hashes of the committed kernels: a live-range / scheduling rewrite must not move a bit).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python scripts/exp/step_bits.py > gpurun_out/bits_new.txt 2>&1 || { tail -3 gpurun_out/bits_new.txt; exit 1; }
SAGA_NN_VALU_PASS=1 timeout -k 10 300 python scripts/exp/step_bits.py > gpurun_out/bits_new_valu.txt 2>&1 || exit 1
diff scripts/exp/step_bits_reference.txt gpurun_out/bits_new.txt && echo BITS_IDENTICAL_TO_REFERENCE
diff gpurun_out/bits_new.txt gpurun_out/bits_new_valu.txt && echo VALU_IDENTICAL
if [ -n "$REF_LIB" ]; then
  SAGA_HIP_LIB=$REF_LIB timeout -k 10 300 python scripts/exp/step_bits.py > gpurun_out/bits_ref.txt 2>&1 && diff gpurun_out/bits_ref.txt gpurun_out/bits_new.txt && echo BITS_IDENTICAL_TO_REF_LIB
fi
tail -3 gpurun_out/bits_new.txt
