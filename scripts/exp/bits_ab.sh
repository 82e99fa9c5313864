set -o pipefail
mkdir -p gpurun_out
SAGA_HIP_LIB=${REF_LIB:-build/variants/libsaga_ref.so} timeout -k 10 300 python scripts/exp/step_bits.py > gpurun_out/bits_ref.txt 2>&1 && \
timeout -k 10 300 python scripts/exp/step_bits.py > gpurun_out/bits_new.txt 2>&1 && \
SAGA_NN_VALU_PASS=1 timeout -k 10 300 python scripts/exp/step_bits.py > gpurun_out/bits_new_valu.txt 2>&1 ; \
diff gpurun_out/bits_ref.txt gpurun_out/bits_new.txt && echo BITS_IDENTICAL; diff gpurun_out/bits_new.txt gpurun_out/bits_new_valu.txt && echo VALU_IDENTICAL; tail -3 gpurun_out/bits_new.txt
