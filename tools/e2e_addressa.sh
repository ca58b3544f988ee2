#!/bin/bash
# The reference README's three Addressa commands, end to end on the GPU box (data/addressa ships with the repo):
#   logs -> gpurun_out/runs/*.log   (copy into profiles/rNN_runs/)
mkdir -p gpurun_out/runs
R=$PWD
( time timeout 600 python ./macr_mf/train.py --dataset addressa --batch_size 1024 --cuda 0 --saveID 1 --log_interval 10 --lr 0.001 --train normalbce --test normal ) > gpurun_out/runs/mf_addressa_normal.log 2>&1
( time timeout 600 python ./macr_mf/train.py --dataset addressa --batch_size 1024 --cuda 0 --saveID 0 --log_interval 10 --lr 0.001 --check_c 1 --c 40 --train rubibceboth --test rubi --alpha 1e-3 --beta 1e-3 ) > gpurun_out/runs/mf_addressa_rubi.log 2>&1
( time timeout 1200 python macr_lightgcn/LightGCN.py --data_path data/ --dataset addressa --verbose 1 --layer_size [64,64] --Ks [20] --loss bceboth --test rubiboth --c 40 --epoch 2000 --early_stop 1 --lr 0.001 --batch_size 1024 --gpu_id 0 --log_interval 10 --alpha 1e-2 --beta 1e-3 ) > gpurun_out/runs/lgcn_addressa_rubi.log 2>&1
tail -n 6 gpurun_out/runs/*.log
