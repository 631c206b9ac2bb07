#!/usr/bin/env python3
"""GIST with an ultra-wide hidden layer (BASELINE config 5: n_hidden 32768 over 8 sub-GCNs) --
CLI of the reference's cluster_gcn/cluster_gcn_ist_ultra_wide.py (same flags and result lines as
cluster_gcn_ist_distrib.py, :573-616).

What the reference changes for this script, and what that becomes on MI355X:

  reference (cluster_gcn_ist_ultra_wide.py)                    here
  ----------------------------------------------------------  -----------------------------------
  base model never leaves host RAM (:84); every weight slice   the base model (8.8 GB at H=32768,
  is fancy-indexed on the host -- an [H, 2h] slab copy per     L=2) is resident in every GPU's HBM;
  site per layer -- moved .to(device), broadcast to one peer   a sync is one RCCL all-gather of the
  inside a fresh 2-rank group, and moved back .to('cpu')       flat sub-model arenas + on-device
  (:111-133,143-156,189-203,224-312)                           block scatters; a dispatch is a
                                                               local block gather (gist_amd/ist.py)
  full-graph evaluation on the CPU, g.cpu() (:500-504),        evaluation stays on the GPU, one
  because [N, 2H] activations do not fit its GPU               block of rows at a time
                                                               (trainer.FullGraphEvaluator)

So there is no second implementation: this entry point runs the same wrapper and loop as
cluster_gcn_ist_distrib with the evaluator's row block bounded for 2H = 65536-wide operands.

    for i in 0 1 2 3 4 5 6 7; do
      python -m gist_amd.scripts.cluster_gcn_ist_ultra_wide --num_subnet 8 --rank $i --cuda-id $i \\
          --n-hidden 32768 --n-layers 2 --iter_per_site 100 --use_layernorm True \\
          --dropout 0.2 --lr 0.01 --n-epochs 40 --dataset reddit-synth &
    done; wait
"""
from gist_amd.scripts.cluster_gcn_ist_distrib import build_parser, main as _main


def main(args=None, dataset=None, log=print):
    return _main(args, dataset=dataset, log=log, ultra_wide=True)


if __name__ == '__main__':
    import os
    os.environ.setdefault('GIST_GC_FREEZE', '1')      # this process is the application: sampler.freeze_setup_objects
    main()
