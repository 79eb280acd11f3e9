cd $GRAFT_REPO_ROOT
python -c "import torch; x=torch.ones(64,11); x[:,0]=torch.linspace(-3,3,64); torch.save(x,'/tmp/init_f.pt')"
export HTA_CPU_BASELINE_TIMING=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 HIP_VISIBLE_DEVICES=
t0=$(date +%s.%N)
for i in $(seq 0 15); do python oracle/cpu_baseline.py funnel-hmc $((1000+i)) 3.0 /tmp/init_f.pt $i 3 > /tmp/w$i.out 2> /tmp/w$i.err & done
wait
t1=$(date +%s.%N); echo "wall $(echo "$t1 - $t0" | bc)"
cat /tmp/w0.err; echo ...; cat /tmp/w7.err | tail -4; wc -c /tmp/w0.out
