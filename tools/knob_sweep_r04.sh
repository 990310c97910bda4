#!/bin/bash
# re-sweep of the task-queue plan's knobs on config M (three-run fits) after the round-4 changes of the chain
base=$(timeout -k 10 200 python3 tools/fit_rate.py 5 2>&1 | grep fits | cut -c1-40)
echo "defaults: $base"
for kv in HBEGP_DAG_RL_NEAR=2 HBEGP_DAG_RL_GROUP=16 HBEGP_DAG_SMALLH=2 HBEGP_DAG_SMALLH=8 HBEGP_DAG_CRIT=0 HBEGP_DAG_CRIT=2 HBEGP_DAG_LAUUM_SPLIT=1 HBEGP_DAG_FINE=0 HBEGP_MAX_CONCURRENT=2 HBEGP_DAG_ADAPT=0 HBEGP_DAG_ORDER_WG=112 HBEGP_DAG_ORDER_WG=128 HBEGP_DAG_ORDER_WG=80; do
  echo "$kv: $(env $kv timeout -k 10 200 python3 tools/fit_rate.py 5 2>&1 | grep fits | cut -c1-40)"
done
echo "defaults again: $(timeout -k 10 200 python3 tools/fit_rate.py 5 2>&1 | grep fits | cut -c1-40)"
