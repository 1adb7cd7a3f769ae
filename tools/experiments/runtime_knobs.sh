# HIP / ROCclr runtime knobs (environment, read at runtime initialisation) against the bf16 train step of bench.py on one box:
# the step has ~770 launches and one host synchronisation, so dispatch-side settings can matter (HIP_FORCE_DEV_KERNARG=0
# costs 1.2 ms: ab_dev_kernarg.sh).  Baseline first and last.
run() {
  echo "$1"
  env $1 timeout 150 python bench.py --mode train --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d.get('train',d)
        print('   ms', round(d['ms_per_step'],3), 'median', round(t.get('step_ms_median'),3), 'slack', round(t.get('host_slack_at_sync_ms'),2), 'loss', t.get('loss_last_timed', t.get('loss')))
"
}
run "BRCNN_NOP=1"
# (first call) run "ROC_ACTIVE_WAIT_TIMEOUT=0"
# (first call) run "ROC_ACTIVE_WAIT_TIMEOUT=100"
# (first call) run "ROC_ACTIVE_WAIT_TIMEOUT=20000"
# (first call) run "ROC_CPU_WAIT_FOR_SIGNAL=0"
# ROC_SYSTEM_SCOPE_SIGNAL=0: the bench never finished (killed by the 1800 s limit of the call) -- not run again
run "AMD_OPT_FLUSH=0"
run "ROC_SKIP_KERNEL_ARG_COPY=1"
run "DEBUG_HIP_KERNARG_COPY_OPT=0"
run "ROC_USE_FGS_KERNARG=0"
run "ROC_AQL_QUEUE_SIZE=65536"
run "ROC_SIGNAL_POOL_SIZE=4096"
run "DEBUG_CLR_MAX_BATCH_SIZE=4096"
run "GPU_MAX_HW_QUEUES=4"
run "GPU_MAX_HW_QUEUES=16"
run "HSA_KERNARG_POOL_SIZE=16777216"
run "BRCNN_NOP=1"
