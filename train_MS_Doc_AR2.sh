#!/bin/bash
# MS-MARCO Document AR2/SimANS iteration on the MI355X engine -- same flags as SimANS/train_MS_Doc_AR2.sh
# (train half; the generate half is simxns_amd/co_training/co_training_generate.py with the RobertaDot embedder).
EXP_NAME=co_training_MS_MARCO_Doc_SimANS
Iteration_step=5000
Iteration_reranker_step=1000
MAX_STEPS=40000
NGPU=${NGPU:-8}
for global_step in `seq 0 $Iteration_step $MAX_STEPS`; do
    python -u -m torch.distributed.run --nproc_per_node=$NGPU --master-addr 127.0.0.1 --master_port=9539 \
    simxns_amd/Doc_training/co_training_doc_train.py \
    --model_type=ckpt/MS-Doc/adore-star \
    --model_name_or_path=ckpt/MS-Doc/checkpoint-20000 \
    --max_seq_length=512 --per_gpu_train_batch_size=32 --gradient_accumulation_steps=1 \
    --number_neg=15 --learning_rate=5e-6 \
    --teacher_model_type=roberta-base \
    --teacher_model_path=ckpt/MS-Doc/checkpoint-reranker20000 \
    --teacher_learning_rate=1e-6 \
    --output_dir=ckpt/$EXP_NAME \
    --log_dir=tensorboard/logs/$EXP_NAME \
    --origin_data_dir=data/MS-Doc/train_ce_0.tsv \
    --train_qa_path=data/MS-Doc/msmarco-doctrain-queries.tsv \
    --passage_path=data/MS-Doc \
    --logging_steps=100 --save_steps=5000 --max_steps=$MAX_STEPS \
    --gradient_checkpointing --distill_loss --fp16 \
    --iteration_step=$Iteration_step \
    --iteration_reranker_step=$Iteration_reranker_step \
    --temperature_distill=1 --ann_dir=ckpt/$EXP_NAME/temp --adv_lambda 1 --global_step=$global_step
done
