# Runs on the GPU box: every fuzz script once with the seed given (each against its exact / float64 counterpart); the output
# ends with one line per script
set -o pipefail
S=${1:-101}
mkdir -p gpurun_out
for job in "nn_fuzz.py 600" "loss_fuzz.py 300" "p2p_fuzz.py 200" "step_fuzz.py 80" "motion_step_fuzz.py 60"; do
  set -- $job
  timeout -k 10 500 python -u scripts/exp/$1 $2 $S > gpurun_out/fuzz_${1%.py}_$S.log 2>&1
  echo "$1 cases $2 seed $S rc=$? : $(grep -c -i mismatch gpurun_out/fuzz_${1%.py}_$S.log) lines with 'mismatch'; last: $(tail -1 gpurun_out/fuzz_${1%.py}_$S.log | cut -c1-120)"
done
