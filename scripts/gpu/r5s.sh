export TMPDIR=/tmp
mkdir -p gpurun_out/r5s
timeout 600 python -c "
import __graft_entry__ as g
g.smoke()
print('smoke ok')
" > gpurun_out/r5s/smoke.log 2>&1; tail -3 gpurun_out/r5s/smoke.log
