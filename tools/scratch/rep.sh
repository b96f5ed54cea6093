# usage: rep.sh "<env assignments>" n
for i in $(seq 1 $2); do env $1 AB_TENSORS=1 python tools/ab_grads.py CGEN_DUMMY 0 1 2>&1 | grep "gradient buffers" | cut -c1-110; done
