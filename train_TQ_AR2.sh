#!/bin/bash
# AR2 + SimANS on TriviaQA -- same entrypoint name, loop and flags as SimANS/train_TQ_AR2.sh (train job on the MI355X engine;
# the generate job is not part of this engine yet, see train_MS_Pas_AR2.sh).
EXP_NAME=co_training_tq_SimANS_test
TB_DIR=tensorboard_log/$EXP_NAME
OUT_DIR=output/$EXP_NAME
DE_CKPT_PATH=ckpt/TQ/triviaqa_fintinue.pkl
CE_CKPT_PATH=ckpt/TQ/checkpoint-reranker34000
Origin_Data_Dir=data/TQ/train_ce_0.json
Iteration_step=2000
Iteration_reranker_step=500
MAX_STEPS=10000
NPROC=${NPROC:-8}
for global_step in `seq 0 $Iteration_step $MAX_STEPS`;
do
    python -u -m torch.distributed.run --nproc_per_node=$NPROC --master-addr 127.0.0.1 --master_port=9539 \
    simxns_amd/wiki/co_training_wiki_train.py \
    --model_type=nghuyong/ernie-2.0-base-en \
    --model_name_or_path=$DE_CKPT_PATH \
    --max_seq_length=256 --per_gpu_train_batch_size=8 --gradient_accumulation_steps=1 \
    --number_neg=15 --learning_rate=5e-6 \
    --reranker_model_type=nghuyong/ernie-2.0-large-en \
    --reranker_model_path=$CE_CKPT_PATH \
    --reranker_learning_rate=1e-6 \
    --output_dir=$OUT_DIR \
    --log_dir=$TB_DIR \
    --origin_data_dir=$Origin_Data_Dir \
    --warmup_steps=1000 --logging_steps=100 --save_steps=2000 --max_steps=$MAX_STEPS \
    --gradient_checkpointing --normal_loss \
    --iteration_step=$Iteration_step \
    --iteration_reranker_step=$Iteration_reranker_step \
    --temperature_normal=1 --ann_dir=$OUT_DIR/temp --adv_lambda 0.0 --global_step=$global_step --a 0.5 --b 0
    g_global_step=`expr $global_step + $Iteration_step`
    if [ -n "$SIMX_GENERATE_CMD" ]; then $SIMX_GENERATE_CMD --global_step=$g_global_step; else
        echo "generate job for step $g_global_step: not provided by simxns_amd (set SIMX_GENERATE_CMD)"; break; fi
done
