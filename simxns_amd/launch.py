"""Iteration driver of the AR2 / SimANS recipes on the MI355X engine.

The reference drives every recipe with a shell loop (SimANS/train_MS_Pas_AR2.sh, train_NQ_AR2.sh, train_TQ_AR2.sh,
train_MS_Doc_AR2.sh): for global_step in 0, iteration_step, ... max_steps: one `train` job (retriever + reranker
alternation, co_training_*_train.py) followed by one `generate` job (re-encode the corpus with the new retriever and mine
the next round's hard negatives).  Here the loop is this module and the hyper-parameters of record live in RECIPES; the
repo-root train_*_AR2.sh files only name a recipe.

    python -m simxns_amd.launch MS_Pas [--nproc 8] [--dry-run] [--first-step N] [--last-step M] [-- extra train flags]
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ms_pas():
    exp = "co_training_MS_MARCO_Pas_SimANS"
    common = dict(model_type="Luyu/co-condenser-marco", max_seq_length=128, output_dir="ckpt/" + exp,
                  log_dir="tensorboard/logs/" + exp, train_qa_path="data/MS-Pas/train.query.txt",
                  dev_qa_path="data/MS-Pas/dev.query.txt", passage_path="data/MS-Pas", gradient_checkpointing=True,
                  ann_dir="ckpt/%s/temp" % exp)
    train = dict(common, model_name_or_path="ckpt/MS-Pas/checkpoint-20000", per_gpu_train_batch_size=16,
                 gradient_accumulation_steps=2, number_neg=15, learning_rate=5e-6,
                 teacher_model_type="nghuyong/ernie-2.0-large-en", teacher_model_path="ckpt/MS-Pas/checkpoint-reranker20000",
                 teacher_learning_rate=5e-7, origin_data_dir="data/MS-Pas/train_ce_0.tsv", logging_steps=10, save_steps=5000,
                 distill_loss=True, temperature_distill=1, adv_lambda=1,
                 sampler="gpu")       # north_star config 2: SimANS draw + collate on the device (--sampler host replays CPython's picks)
    return dict(iteration_step=5000, iteration_reranker_step=500, max_steps=60000,
                train=("simxns_amd/co_training/co_training_marco_train.py", train),
                generate=("simxns_amd/co_training/co_training_generate.py", dict(common, adv_step=0)))


def _wiki(name, exp, de_ckpt, ce_ckpt, data, max_steps, seq_len, lr, warmup, extra, qa):
    out = "output/" + exp
    train = dict(model_type="nghuyong/ernie-2.0-base-en", model_name_or_path=de_ckpt, max_seq_length=seq_len,
                 per_gpu_train_batch_size=8, gradient_accumulation_steps=1, number_neg=15, learning_rate=lr,
                 reranker_model_type="nghuyong/ernie-2.0-large-en", reranker_model_path=ce_ckpt, reranker_learning_rate=1e-6,
                 output_dir=out, log_dir="tensorboard_log/" + exp, origin_data_dir=data, warmup_steps=warmup, logging_steps=100,
                 save_steps=2000, gradient_checkpointing=True, normal_loss=True, temperature_normal=1, ann_dir=out + "/temp")
    train.update(extra)
    # second command of the round (train_NQ_AR2.sh:34-52): re-embed psgs_w100.tsv, mine top-100, write <ann_dir>/train_ce_<step>.json
    generate = dict(model_type="nghuyong/ernie-2.0-base-en", model_name_or_path=de_ckpt, max_seq_length=seq_len,
                    per_gpu_train_batch_size=8, output_dir=out, log_dir="tensorboard/logs/" + exp, origin_data_dir=data,
                    origin_data_dir_dev=data.replace("train_ce_0", "dev_ce_0"), train_qa_path=qa % "train", test_qa_path=qa % "test",
                    dev_qa_path=qa % "dev", passage_path="data/psgs_w100.tsv", gradient_checkpointing=True, ann_dir=out + "/temp")
    return dict(iteration_step=2000, iteration_reranker_step=500, max_steps=max_steps,
                train=("simxns_amd/wiki/co_training_wiki_train.py", train),
                generate=("simxns_amd/wiki/co_training_wiki_generate.py", generate))


def _ms_doc():
    exp = "co_training_MS_MARCO_Doc_SimANS"
    train = dict(model_type="ckpt/MS-Doc/adore-star", model_name_or_path="ckpt/MS-Doc/checkpoint-20000", max_seq_length=512,
                 per_gpu_train_batch_size=32, gradient_accumulation_steps=1, number_neg=15, learning_rate=5e-6,
                 teacher_model_type="roberta-base", teacher_model_path="ckpt/MS-Doc/checkpoint-reranker20000",
                 teacher_learning_rate=1e-6, output_dir="ckpt/" + exp, log_dir="tensorboard/logs/" + exp,
                 origin_data_dir="data/MS-Doc/train_ce_0.tsv", train_qa_path="data/MS-Doc/msmarco-doctrain-queries.tsv",
                 passage_path="data/MS-Doc", logging_steps=100, save_steps=5000, gradient_checkpointing=True, distill_loss=True,
                 temperature_distill=1, ann_dir="ckpt/%s/temp" % exp, adv_lambda=1)      # (no --fp16, as train_MS_Doc_AR2.sh:9-26;
                                                                                         #  SIMX_DTYPE=fp16 opts any recipe in)
    generate = dict(model_type="ckpt/MS-Doc/adore-star", max_seq_length=512, output_dir="ckpt/" + exp,
                    log_dir="tensorboard/logs/" + exp, train_qa_path="data/MS-Doc/msmarco-doctrain-queries.tsv",
                    dev_qa_path="data/MS-Doc/msmarco-docdev-queries.tsv", passage_path="data/MS-Doc", gradient_checkpointing=True,
                    ann_dir="ckpt/%s/temp" % exp)                                        # train_MS_Doc_AR2.sh:28-39
    return dict(iteration_step=5000, iteration_reranker_step=1000, max_steps=40000,
                train=("simxns_amd/Doc_training/co_training_doc_train.py", train),
                generate=("simxns_amd/Doc_training/co_training_doc_generate.py", generate))


RECIPES = {
    "MS_Pas": _ms_pas,
    "NQ": lambda: _wiki("NQ", "co_training_nq_SimANS_test", "ckpt/NQ/nq_fintinue.pkl", "ckpt/NQ/checkpoint-reranker26000",
                        "data/NQ/train_ce_0.json", 30000, 128, 1e-5, 2000, dict(adv_lambda=0, b=1.0), "data/NQ/nq-%s.qa.csv"),
    "TQ": lambda: _wiki("TQ", "co_training_tq_SimANS_test", "ckpt/TQ/triviaqa_fintinue.pkl", "ckpt/TQ/checkpoint-reranker34000",
                        "data/TQ/train_ce_0.json", 10000, 256, 5e-6, 1000, dict(adv_lambda=0.0, a=0.5, b=0), "data/TQ/trivia-%s.qa.csv"),
    "MS_Doc": _ms_doc,
}


def job_argv(script, flags, nproc, port):
    argv = [sys.executable, "-u", "-m", "torch.distributed.run", "--nproc_per_node=%d" % nproc, "--master-addr", "127.0.0.1",
            "--master_port=%d" % port, script]
    for k, v in flags.items():
        if v is True:
            argv.append("--" + k)
        elif v is not False and v is not None:
            argv.append("--%s=%s" % (k, v))
    return argv


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("recipe", choices=sorted(RECIPES))
    ap.add_argument("--nproc", type=int, default=int(os.environ.get("NPROC", os.environ.get("NGPU", "8"))))
    ap.add_argument("--port", type=int, default=9539)
    ap.add_argument("--first-step", type=int, default=0)
    ap.add_argument("--last-step", type=int, default=None)
    ap.add_argument("--dry-run", action="store_true", help="print the job command lines instead of running them")
    ap.add_argument("--fold-accumulation", action="store_true",
                    help="run the gradient_accumulation_steps micro-batches of an optimizer step as ONE batch (MS_Pas: 2 x 16 queries -> "
                         "32).  The recipes split the step for 32-40 GB GPUs; their losses are per-query (no in-batch negatives), so the "
                         "summed gradient is the same up to rounding and dropout draws, and 65536 passage tokens fill the 256 CUs in "
                         "whole waves of GEMM tiles where 32768 leave the last wave half empty.  Off by default: the recipe to the letter")
    args, extra = ap.parse_known_args(argv)
    extra = [e for e in extra if e != "--"]
    r = RECIPES[args.recipe]()
    if args.fold_accumulation:
        flags = r["train"][1]
        if flags.get("inbatch") or flags.get("in_batch"):
            ap.error("--fold-accumulation: this recipe scores in-batch negatives; a larger batch changes its loss")
        flags["per_gpu_train_batch_size"] *= flags["gradient_accumulation_steps"]
        flags["gradient_accumulation_steps"] = 1
    step, last = r["iteration_step"], r["max_steps"] if args.last_step is None else args.last_step
    for global_step in range(args.first_step, last + 1, step):
        loop = dict(max_steps=r["max_steps"], iteration_step=step, iteration_reranker_step=r["iteration_reranker_step"])
        script, flags = r["train"]
        jobs = [job_argv(script, dict(flags, global_step=global_step, **loop), args.nproc, args.port) + extra]
        if r["generate"] is not None:
            script, flags = r["generate"]
            jobs.append(job_argv(script, dict(flags, global_step=global_step + step, **loop), args.nproc, args.port))
        elif os.environ.get("SIMX_GENERATE_CMD"):
            jobs.append(os.environ["SIMX_GENERATE_CMD"].split() + ["--global_step=%d" % (global_step + step)])
        for j in jobs:
            if args.dry_run:
                print(" ".join(j))
            else:
                subprocess.check_call(j, cwd=ROOT)
        if r["generate"] is None and not os.environ.get("SIMX_GENERATE_CMD"):
            print("recipe %s: no generate job in this engine (set SIMX_GENERATE_CMD); stopping after the first train job" % args.recipe)
            break
    return 0


if __name__ == "__main__":
    sys.exit(main())
