# Helper for gpurun command lines: run_step NAME SECONDS cmd... runs one GPU step under `timeout -k 10`, logs to
# gpurun_out/$OUT/NAME.log and STOPS the whole script if the step was killed by its time limit (no GPU step may follow a
# hung one); an ordinary non-zero exit is recorded and the script goes on.
run_step() {
  local name=$1 secs=$2; shift 2
  echo "== $name: $*" | tee -a gpurun_out/$OUT/steps.txt
  timeout -k 10 $secs "$@" > gpurun_out/$OUT/$name.log 2> gpurun_out/$OUT/$name.err
  local rc=$?
  echo "   rc=$rc" | tee -a gpurun_out/$OUT/steps.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name hit its time limit: stopping" | tee -a gpurun_out/$OUT/steps.txt; exit 99; fi
  return 0
}
