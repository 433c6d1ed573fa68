# Helper for gpurun command lines: run_step NAME SECONDS cmd... runs one GPU step in its own process group under
# `timeout -k 10`, logs to gpurun_out/$OUT/NAME.log / NAME.err and STOPS the whole script (exit 99) if the step
#  * was killed by its time limit (no GPU step may follow a hung one), or
#  * reported a GPU fault: a process that died of a queue abort (signal 6, "HSA_STATUS_ERROR_*", "Memory access fault") can
#    leave its wrapper (rocprofv3, pytest's parent) waiting until the limit - the step's output is polled every 5 s and the
#    process group is ended as soon as such a line appears, instead of sitting out the limit (round 3, r3j: 300 s lost).
# An ordinary non-zero exit is recorded and the script goes on.
run_step() {
  local name=$1 secs=$2; shift 2
  echo "== $name: $*" | tee -a gpurun_out/$OUT/steps.txt
  local log=gpurun_out/$OUT/$name.log err=gpurun_out/$OUT/$name.err
  setsid timeout -k 10 $secs "$@" > $log 2> $err &
  local pid=$! fault=0
  while kill -0 $pid 2>/dev/null; do
    sleep 5
    if grep -qE "HSA_STATUS_ERROR|Memory access fault|Aborted \(core dumped\)|hipErrorLaunchFailure|hipErrorIllegalAddress" $err $log 2>/dev/null; then
      fault=1
      sleep 5                                  # let the dying process finish its message
      kill -TERM -- -$pid 2>/dev/null; sleep 2; kill -KILL -- -$pid 2>/dev/null
      break
    fi
  done
  wait $pid; local rc=$?
  echo "   rc=$rc" | tee -a gpurun_out/$OUT/steps.txt
  if [ $fault -eq 1 ]; then echo "step $name reported a GPU fault: stopping (no GPU step after a fault)" | tee -a gpurun_out/$OUT/steps.txt; exit 99; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name hit its time limit: stopping" | tee -a gpurun_out/$OUT/steps.txt; exit 99; fi
  return 0
}
