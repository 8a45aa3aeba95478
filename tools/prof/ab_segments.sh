P=tools/pipeline_bench/_build/pipeline_bench
for s in 5 4 6 7 3; do
  echo -n "SEGMENT_LOG2=$s: "
  BLITZAR_AMD_SEGMENT_LOG2=$s $P --steps 100 --warmup 10 | grep '^{' | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('seq %.4f lone %.4f acc_seq %.4f lone stages %s'%(d['ms_per_step'], d['lone_ms'], d['acc_in_sequence_ms'], d['lone_stage_ms']))"
done
for l in 17 18 19 21; do
  echo -n "log2n=$l: "
  $P --steps 100 --warmup 10 --log2n $l | grep '^{' | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('seq %.4f lone %.4f acc_seq %.4f lone stages %s'%(d['ms_per_step'], d['lone_ms'], d['acc_in_sequence_ms'], d['lone_stage_ms']))"
done
