"""Train job of the NQ / TriviaQA AR2+SimANS iteration on the MI355X engine -- CLI, phase machine and loss of
SimANS/wiki/co_training_wiki_train.py (train() :85-330, flags :380-583): starts in the reranker phase (:127), loss =
adv_lambda * sum(reward * log(p+1e-7)) + (1-adv_lambda) * (-sum(p_T * log(p+1e-7)) / B) (:203-228), TraditionDataset with
the Gaussian SimANS sampler (--a/--b), DistributedSampler, ``train_ce_<step>.json`` files.
Shares the runtime pieces with co_training/co_training_marco_train.py."""
import json
import logging
import os
import sys

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, RandomSampler
from torch.utils.data.distributed import DistributedSampler

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simxns_amd import ops                                                                  # noqa: E402
from simxns_amd.co_training import co_training_marco_train as M                              # noqa: E402
from simxns_amd.model.models import BiBertEncoder, HFBertEncoder, Reranker                   # noqa: E402
from simxns_amd.optim import LinearWarmupSchedule                                            # noqa: E402
from simxns_amd.utils.MARCO_until_new import HashTokenizer                                   # noqa: E402
from simxns_amd.utils.dpr_utils import load_states_from_checkpoint                           # noqa: E402
from simxns_amd.utils.util_wiki import TraditionDataset, is_first_worker, set_seed           # noqa: E402

logger = logging.getLogger(__name__)


def get_arguments(argv=None):
    import argparse
    pre = argparse.ArgumentParser(add_help=False)
    for f, t, d in (("--reranker_model_path", str, ""), ("--reranker_model_type", str, ""), ("--reranker_learning_rate", float, 0),
                    ("--temperature_normal", float, 1), ("--a", float, 0.5), ("--b", float, 0)):
        pre.add_argument(f, type=t, default=d)
    pre.add_argument("--normal_loss", default=False, action="store_true")
    pre.add_argument("--normal_term", type=str, default="cross_e")
    wiki, rest = pre.parse_known_args(argv)
    args = M.get_arguments(rest)
    for k, v in vars(wiki).items():
        setattr(args, k, v)
    args.teacher_model_type = args.reranker_model_type or args.teacher_model_type
    args.teacher_model_path = args.reranker_model_path or args.teacher_model_path
    args.teacher_learning_rate = args.reranker_learning_rate or args.teacher_learning_rate
    return args


def train(args, model, reranker_model, tokenizer, global_step=0):
    model.to(args.device)
    reranker_model.to(args.device)
    args.train_batch_size = args.per_gpu_train_batch_size * max(1, args.n_gpu)
    optimizer = M.get_optimizer(args, model, args.weight_decay, args.learning_rate, args.adam_epsilon)
    reranker_optimizer = M.get_optimizer(args, reranker_model, args.weight_decay, args.teacher_learning_rate, args.adam_epsilon)
    world = dist.get_world_size() if args.local_rank != -1 else 1
    if world > 1:
        optimizer.enable_overlap(world)
        reranker_optimizer.enable_overlap(world)
    tr_loss = tr_normal = tr_contr = 0.0
    model.zero_grad()
    reranker_model.zero_grad()
    set_seed(args)
    train_flag, step = 1, 0                                   # starts with the reranker (:127)
    r_max = args.max_steps * (1 - args.iteration_reranker_step / args.iteration_step)
    k_max = args.max_steps * (args.iteration_reranker_step / args.iteration_step)
    scheduler = LinearWarmupSchedule(optimizer, 0.1 * r_max, r_max)
    reranker_scheduler = LinearWarmupSchedule(reranker_optimizer, 0.1 * k_max, k_max)
    if global_step != 0:
        path = os.path.join(args.ann_dir, 'train_ce_' + str(global_step) + '.json')
        M._load_saved_state(model, optimizer, scheduler, load_states_from_checkpoint(os.path.join(args.output_dir, 'checkpoint-' + str(global_step))))
        M._load_saved_state(reranker_model, reranker_optimizer, reranker_scheduler,
                            load_states_from_checkpoint(os.path.join(args.output_dir, 'checkpoint-reranker' + str(global_step))))
    else:
        path = args.origin_data_dir
    ds = TraditionDataset(path, tokenizer, num_hard_negatives=args.number_neg, a=args.a, b=args.b,
                          max_seq_length=args.max_seq_length, max_q_length=args.max_query_length)
    sampler = RandomSampler(ds) if args.local_rank == -1 else DistributedSampler(ds)
    dl = DataLoader(ds, sampler=sampler, collate_fn=TraditionDataset.get_collate_fn(args), batch_size=args.train_batch_size,
                    num_workers=min(args.num_workers, 10))
    it = iter(dl)
    while global_step < args.max_steps:
        try:
            batch = next(it)
        except StopIteration:
            it = iter(dl)
            batch = next(it)
            if world > 1:
                dist.barrier()
        step += 1
        optimizer.armed = reranker_optimizer.armed = (step + 1) % args.gradient_accumulation_steps == 0
        q_ids, q_mask, c_ids, c_mask = (t.long().to(args.device) for t in batch['retriever'][:4])
        t_ids, t_mask = (t.long().to(args.device) for t in batch['reranker'][:2])
        if train_flag == 0:
            model.train()
            reranker_model.eval()
            q, c = model(q_ids, q_mask, c_ids, c_mask)
            with torch.no_grad():
                z = reranker_model(t_ids, t_mask)
            loss, normal_loss, adv_loss, _ = ops.wiki_normal_adv_loss(q, c, z, args.temperature_normal, args.adv_lambda,
                                                                      args.scale_simmila, args.gradient_accumulation_steps)
            loss.backward()
            tr_loss += loss.item()
            tr_normal += normal_loss.item()
        if train_flag == 1:
            reranker_model.train()
            model.eval()
            z = reranker_model(t_ids, t_mask)
            loss, contr = ops.teacher_ce_loss(z, args.gradient_accumulation_steps)
            loss.backward()
            tr_loss += loss.item()
            tr_contr += contr.item()
        if (step + 1) % args.gradient_accumulation_steps == 0:
            if train_flag == 0:                       # optimizer first, scheduler second (co_training_wiki_train.py:252-254)
                optimizer.step(max_grad_norm=args.max_grad_norm, world_size=world)
                scheduler.step()
            else:
                reranker_optimizer.step(max_grad_norm=args.max_grad_norm, world_size=world)
                reranker_scheduler.step()
            global_step += 1
            if args.logging_steps > 0 and global_step % args.logging_steps == 0:
                logs = {"learning_rate": scheduler.get_last_lr()[0], "loss": tr_loss / args.logging_steps,
                        "normal_loss": tr_normal / args.logging_steps, "contr_loss": tr_contr / args.logging_steps}
                tr_loss = tr_normal = tr_contr = 0.0
                if is_first_worker():
                    logger.info(json.dumps({**logs, **{"step": global_step}}))
            r = global_step % args.iteration_step
            if r > args.iteration_reranker_step:
                train_flag = 0
            elif 0 < r < args.iteration_reranker_step:
                train_flag = 1
            elif r == 0:
                if is_first_worker():
                    M._save_checkpoint(args, model, optimizer, scheduler, global_step)
                    M._save_checkpoint(args, reranker_model, reranker_optimizer, reranker_scheduler, global_step, "checkpoint-reranker")
                if world > 1:
                    dist.barrier()
                train_flag = 1
                break
            if global_step >= args.max_steps:
                break
    return global_step


def main(argv=None):
    args = get_arguments(argv)
    M.set_env(args)
    tokenizer, model, reranker = M.load_model(args)
    if args.output_dir and is_first_worker():
        os.makedirs(args.output_dir, exist_ok=True)
    if args.local_rank != -1:
        dist.barrier()
    return train(args, model, reranker, tokenizer, args.global_step)


if __name__ == "__main__":
    main()
