"""Train job of the AR2/SimANS iteration on the MI355X engine -- same CLI, phase machine, checkpoint files and
log lines as SimANS/co_training/co_training_marco_train.py (flags :361-563, train() :82-307, checkpoints :310-358).

Launch exactly like the reference (``train_MS_Pas_AR2.sh``):
    python -m torch.distributed.run --nproc_per_node=8 simxns_amd/co_training/co_training_marco_train.py <same flags>
(``torch.distributed.launch`` with ``--local_rank`` also works.)  Differences, all stated in INTEGRATION.md:
no apex / DDP wrappers (gradients live in flat buffers that FusedAdamW all-reduces over RCCL), ``--fp16`` selects the
fp16 engine with FusedAdamW's dynamic loss scale, tensorboard is optional, ``--tokenizer_name hash`` selects the offline hash tokenizer.
"""
import argparse
import json
import logging
import os
import sys

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, RandomSampler

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simxns_amd import ops                                                           # noqa: E402
from simxns_amd.model.models import BiBertEncoder, HFBertEncoder, Reranker            # noqa: E402
from simxns_amd.optim import FusedAdamW, LinearWarmupSchedule                         # noqa: E402
from simxns_amd.utils.MARCO_until_new import Rocketqa_v2Dataset, HashTokenizer        # noqa: E402
from simxns_amd.utils.dpr_utils import (CheckpointState, get_model_obj,               # noqa: E402
                                        load_states_from_checkpoint, save_checkpoint_state)
from simxns_amd.utils.util import is_first_worker, set_seed                           # noqa: E402

logger = logging.getLogger(__name__)


def get_optimizer(args, model, weight_decay=0.0, lr=0.0, eps=0.0):
    """co_training_marco_train.py:57-69 ('bias' / 'LayerNorm.weight' never decay; only adamW is known)."""
    if args.optimizer == "adamW":
        return FusedAdamW(model, lr=lr, eps=eps, weight_decay=weight_decay)
    raise Exception("optimizer {0} not recognized! Can only be adamW".format(args.optimizer))


def get_bert_reader_components(args, **kwargs):
    encoder = HFBertEncoder.init_encoder(args, model_type=args.teacher_model_type)
    return Reranker(encoder, encoder.config.hidden_size)


def _load_saved_state(model, optimizer, scheduler, saved_state: CheckpointState):
    """co_training_marco_train.py:348-358: model, optimizer and scheduler state are all restored (the reference loads the
    two dicts unconditionally).  Both the reference's torch formats and this package's round-1 flat format are read; a
    state that cannot be applied raises -- nothing is skipped silently."""
    get_model_obj(model).load_state_dict(saved_state.model_dict, strict=False)
    if saved_state.optimizer_dict:
        optimizer.load_state_dict(saved_state.optimizer_dict)
    else:
        logger.warning("checkpoint carries no optimizer state: Adam moments restart from zero")
    if saved_state.scheduler_dict:
        scheduler.load_state_dict(saved_state.scheduler_dict)
    else:
        logger.warning("checkpoint carries no scheduler state: the schedule restarts at step 0")
    return saved_state.offset if isinstance(saved_state.offset, int) else 0


def _save_checkpoint(args, model, optimizer, scheduler, step, name="checkpoint-"):
    cp = os.path.join(args.output_dir, name + str(step))
    save_checkpoint_state(cp, model, optimizer, scheduler, step, 0, None)
    logger.info("Saved checkpoint at %s", cp)
    return cp


def _bi_encode(model, q_ids, q_mask, c_ids, c_mask):
    return model(q_ids, q_mask, c_ids, c_mask)


_SIDE_STREAMS = {}


def _teacher_stream(device):
    if device not in _SIDE_STREAMS:
        _SIDE_STREAMS[device] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[device]


def train(args, model, teacher_model, tokenizer, global_step=0, dataset_cls=Rocketqa_v2Dataset, dataset_kwargs=None,
          encode_pair=_bi_encode):
    """``dataset_cls`` / ``dataset_kwargs`` / ``encode_pair`` let the MS-Doc job (Doc_training/co_training_doc_train.py:
    RobertaDot + Doc_v2Dataset) reuse this loop; the defaults are the MS-Pas job."""
    tb_writer = None
    if is_first_worker():
        try:
            from torch.utils.tensorboard import SummaryWriter
            tb_writer = SummaryWriter(log_dir=args.log_dir)
        except Exception:                                  # tensorboard is not in this image
            tb_writer = None
    model.to(args.device)
    teacher_model.to(args.device)
    args.train_batch_size = args.per_gpu_train_batch_size * max(1, args.n_gpu)
    optimizer = get_optimizer(args, model, args.weight_decay, args.learning_rate, args.adam_epsilon)
    teacher_optimizer = get_optimizer(args, teacher_model, args.weight_decay, args.teacher_learning_rate, args.adam_epsilon)
    world = dist.get_world_size() if args.local_rank != -1 else 1
    if world > 1:          # DDP's role (:107-114): gradient slices are all-reduced over RCCL while the backward still runs
        optimizer.enable_overlap(world)
        teacher_optimizer.enable_overlap(world)

    tr_loss = tr_distll_loss = tr_contr_loss = 0.0
    model.zero_grad()
    teacher_model.zero_grad()
    set_seed(args)
    train_flag, step = 0, 0
    student_max_step = args.max_steps * (1 - args.iteration_reranker_step / args.iteration_step)
    teacher_max_step = args.max_steps * (args.iteration_reranker_step / args.iteration_step)
    scheduler = LinearWarmupSchedule(optimizer, 0.1 * student_max_step, student_max_step)
    teacher_scheduler = LinearWarmupSchedule(teacher_optimizer, 0.1 * teacher_max_step, teacher_max_step)
    if global_step != 0:
        _load_saved_state(model, optimizer, scheduler,
                          load_states_from_checkpoint(os.path.join(args.output_dir, 'checkpoint-' + str(global_step))))
        _load_saved_state(teacher_model, teacher_optimizer, teacher_scheduler,
                          load_states_from_checkpoint(os.path.join(args.output_dir, 'checkpoint-reranker' + str(global_step))))

    def iteration_loader(gstep):
        """Dataset + batch iterator of the shell-loop iteration that starts at optimiser step `gstep` (train_ce_<gstep>.tsv)."""
        train_data_path = os.path.join(args.ann_dir, 'train_ce_' + str(gstep) + '.tsv') if gstep != 0 else args.origin_data_dir
        train_dataset = dataset_cls(train_data_path, tokenizer, num_hard_negatives=args.number_neg,
                                    trainer_id=max(args.local_rank, 0), trainer_num=world,
                                    corpus_path=args.passage_path, rand_pool=100, **(dataset_kwargs or {}))
        gpu_sampler = getattr(args, "sampler", "host") == "gpu" and hasattr(train_dataset, "build_device_pool")
        if gpu_sampler:
            # --sampler gpu: the SimANS draw and the collate run on the device from tables tokenised once per iteration
            # (Rocketqa_v2Dataset.build_device_pool / device_batch); the host only deals out row numbers
            pool = train_dataset.build_device_pool(args.device)
            logger.info("device pool: %d queries, %d passages, %d candidates per query resident in HBM", pool["q_tok"].shape[0],
                        pool["p_tok"].shape[0], pool["cand_rows"].shape[1])

            class _DeviceBatches(object):
                # the Philox draw is keyed by (seed, rank, micro-step OF THE WHOLE RUN): a relaunch at optimiser step gstep
                # continues the counter instead of replaying the first iteration's picks
                n = gstep * args.gradient_accumulation_steps

                def __iter__(self_):
                    order = list(RandomSampler(train_dataset))
                    for lo in range(0, len(order), args.train_batch_size):
                        self_.n += 1
                        yield train_dataset.device_batch(order[lo:lo + args.train_batch_size], seed=args.seed + max(args.local_rank, 0),
                                                         step=self_.n)
            loader = _DeviceBatches()
        else:
            loader = DataLoader(train_dataset, sampler=RandomSampler(train_dataset),
                                collate_fn=dataset_cls.get_collate_fn(args),
                                batch_size=args.train_batch_size, num_workers=args.num_workers)
        return train_dataset, loader
    train_dataset, train_dataloader = iteration_loader(global_step)
    it = iter(train_dataloader)
    logger.info("***** Running training *****  max steps %d, per-GPU batch %d, accumulation %d, examples %d",
                args.max_steps, args.per_gpu_train_batch_size, args.gradient_accumulation_steps, len(train_dataset))
    while global_step < args.max_steps:
        try:
            batch = next(it)
        except StopIteration:
            it = iter(train_dataloader)
            batch = next(it)
            if world > 1:
                dist.barrier()
        step += 1
        # an accumulated (in-place) gradient buffer is reduced once, by the last micro-step's backward
        optimizer.armed = teacher_optimizer.armed = (step + 1) % args.gradient_accumulation_steps == 0
        bs = batch['student']
        q_ids, q_mask, c_ids, c_mask = (t.long().to(args.device) for t in bs[:4])
        t_ids, t_mask = (t.long().to(args.device) for t in batch['teacher'][:2])
        if train_flag == 0:                                       # retriever step: teacher distils the student
            model.train()
            teacher_model.eval()
            if t_ids.is_cuda:
                # the frozen teacher's forward is independent of the student's: its own HIP stream, joined before the loss
                cur = torch.cuda.current_stream()
                side = _teacher_stream(t_ids.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side), torch.no_grad():
                    relevance_logits = teacher_model(t_ids, t_mask)
                local_q_vector, local_ctx_vectors = encode_pair(model, q_ids, q_mask, c_ids, c_mask)
                cur.wait_stream(side)
                relevance_logits.record_stream(cur)
            else:
                local_q_vector, local_ctx_vectors = encode_pair(model, q_ids, q_mask, c_ids, c_mask)
                with torch.no_grad():
                    relevance_logits = teacher_model(t_ids, t_mask)
            # einsum + softmax + KLDivLoss(batchmean)((p+1e-7).log(), softmax(z/T)) / accum : one kernel (:199-217)
            loss, distill_loss, _ = ops.kl_distill_loss(local_q_vector, local_ctx_vectors, relevance_logits,
                                                        args.temperature_distill, args.scale_simmila,
                                                        args.gradient_accumulation_steps)
            loss.backward()
            tr_loss += loss.item()
            tr_distll_loss += distill_loss.item()
        if train_flag == 1:                                       # teacher (reranker) step: CE with target 0 (:225-245)
            teacher_model.train()
            model.eval()
            relevance_logits = teacher_model(t_ids, t_mask)
            loss, contr_loss = ops.teacher_ce_loss(relevance_logits, args.gradient_accumulation_steps)
            loss.backward()
            tr_loss += loss.item()
            tr_contr_loss += contr_loss.item()
        if (step + 1) % args.gradient_accumulation_steps == 0:    # sic: step starts at 1 (:184, :246)
            # optimizer first, scheduler second, as the reference (:250-252 / :258-260): update k uses lr*lambda(k-1)
            if train_flag == 0:
                optimizer.step(max_grad_norm=args.max_grad_norm, world_size=world)     # clip + AdamW + zero_grad
                scheduler.step()
            if train_flag == 1:
                teacher_optimizer.step(max_grad_norm=args.max_grad_norm, world_size=world)
                teacher_scheduler.step()
            global_step += 1
            if args.logging_steps > 0 and global_step % args.logging_steps == 0:
                logs = {"learning_rate": scheduler.get_last_lr()[0], "loss": tr_loss / args.logging_steps,
                        "distill_loss": tr_distll_loss / args.logging_steps, "contr_loss": tr_contr_loss / args.logging_steps}
                tr_loss = tr_distll_loss = tr_contr_loss = 0.0
                if is_first_worker():
                    if tb_writer is not None:
                        for k, v in logs.items():
                            tb_writer.add_scalar(k, v, global_step)
                    logger.info(json.dumps({**logs, **{"step": global_step}}))
            r = global_step % args.iteration_step                   # phase machine (:283-297)
            if r > args.iteration_reranker_step:
                train_flag = 0
            elif 0 < r < args.iteration_reranker_step:
                train_flag = 1
                if global_step / args.iteration_step < 1:
                    train_flag = 0
            elif r == 0:
                if is_first_worker():
                    _save_checkpoint(args, model, optimizer, scheduler, global_step)
                    _save_checkpoint(args, teacher_model, teacher_optimizer, teacher_scheduler, global_step, "checkpoint-reranker")
                if world > 1:
                    dist.barrier()
                train_flag = 0
                if os.environ.get("SIMX_CONTINUE_AT_BOUNDARY") == "1" and global_step < args.max_steps:
                    # test hook (tests/test_train_script_gpu.py): run the next shell-loop iteration IN THIS PROCESS exactly as
                    # a relaunch with --global_step would start it -- reseed, re-read the mined file, restart the counters a
                    # fresh process starts at zero -- but on the live model / optimiser / scheduler objects, i.e. WITHOUT the
                    # checkpoint round trip: the uninterrupted reference a resumed job is compared with
                    set_seed(args)
                    for mod in list(model.modules()) + list(teacher_model.modules()):
                        if getattr(mod, "engine", None) is not None:
                            mod.engine._drop_calls = 0
                    train_dataset, train_dataloader = iteration_loader(global_step)
                    it = iter(train_dataloader)
                    step = 0
                    tr_loss = tr_distll_loss = tr_contr_loss = 0.0
                    continue
                break
            if args.save_steps > 0 and global_step % args.save_steps == 0 and is_first_worker():
                _save_checkpoint(args, model, optimizer, scheduler, global_step)
                _save_checkpoint(args, teacher_model, teacher_optimizer, teacher_scheduler, global_step, "checkpoint-reranker")
            if global_step >= args.max_steps:
                break
    if tb_writer is not None:
        tb_writer.close()
    return global_step


def get_arguments(argv=None):
    p = argparse.ArgumentParser()
    A = p.add_argument
    A("--model_type", default=None, type=str)
    A("--model_name_or_path", default=None, type=str)
    A("--output_dir", default=None, type=str)
    A("--num_epoch", default=0, type=int)
    A("--config_name", default="", type=str)
    A("--tokenizer_name", default="", type=str)
    A("--max_seq_length", default=128, type=int)
    A("--max_query_length", default=32, type=int)
    A("--triplet", default=False, action="store_true")
    A("--log_dir", default=None, type=str)
    A("--optimizer", default="adamW", type=str)
    A("--per_gpu_train_batch_size", default=8, type=int)
    A("--gradient_accumulation_steps", type=int, default=1)
    A("--learning_rate", default=5e-5, type=float)
    A("--weight_decay", default=0.0, type=float)
    A("--adam_epsilon", default=1e-8, type=float)
    A("--max_grad_norm", default=2.0, type=float)
    A("--max_steps", default=300000, type=int)
    A("--warmup_steps", default=0, type=int)
    A("--logging_steps", type=int, default=500)
    A("--save_steps", type=int, default=500)
    A("--no_cuda", action="store_true")
    A("--seed", type=int, default=42)
    A("--sampler", default="host", choices=["host", "gpu"],
      help="gpu: SimANS draw + collate on the device from a pool tokenised once per iteration (no DataLoader workers)")
    A("--fp16", action="store_true")
    A("--fp16_opt_level", type=str, default="O1")
    A("--single_warmup", default=False, action="store_true")
    A("--adv_training", default=False, action="store_true")
    A("--gradient_checkpointing", default=False, action="store_true")
    A("--origin_data_dir", default=None, type=str)
    A("--origin_data_dir_dev", default=None, type=str)
    A("--fix_embedding", default=False, action="store_true")
    A("--continue_train", default=False, action="store_true")
    A("--adv_data_path", type=str, default=None)
    A("--ann_data_path", type=str, default=None)
    A("--distill_loss", default=False, action="store_true")
    A("--share_weight", default=False, action="store_true")
    A("--adv_norm", default=0.3, type=float)
    A("--teacher_model_path", type=str, default="")
    A("--teacher_model_type", type=str, default="")
    A("--number_neg", type=int, default=20)
    A("--adv_lambda", default=0., type=float)
    A("--adv_steps", default=3, type=int)
    A("--local_rank", "--local-rank", dest="local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", -1)))
    A("--server_ip", type=str, default="")
    A("--server_port", type=str, default="")
    A("--test_qa_path", type=str, default="")
    A("--train_qa_path", type=str, default="")
    A("--dev_qa_path", type=str, default="")
    A("--passage_path", type=str, default="")
    A("--iteration_step", default=80, type=int)
    A("--iteration_reranker_step", default=40, type=int)
    A("--temperature_distill", default=3, type=float)
    A("--scale_simmila", default=False, action="store_true")
    A("--teacher_learning_rate", default=0, type=float)
    A("--load_cache", default=False, action="store_true")
    A("--ann_dir", type=str, default="")
    A("--global_step", type=int, default=0)
    A("--num_workers", type=int, default=15)            # reference hard-codes 15 (:156)
    return p.parse_args(argv)


def set_env(args):
    if args.local_rank == -1 or args.no_cuda:
        if not torch.cuda.is_available():
            raise SystemExit("co_training_marco_train: a HIP device is required (the engine has no CPU fallback)")
        device = torch.device("cuda")
        args.n_gpu = 1
    else:
        torch.cuda.set_device(args.local_rank)
        device = torch.device("cuda", args.local_rank)
        dist.init_process_group(backend="nccl")          # RCCL on ROCm
        args.n_gpu = 1
    args.device = device
    args.rank = dist.get_rank() if args.local_rank != -1 else 0
    args.world_size = dist.get_world_size() if args.local_rank != -1 else 1
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s -   %(message)s", datefmt="%m/%d/%Y %H:%M:%S",
                        level=logging.INFO if args.local_rank in [-1, 0] else logging.WARN)
    set_seed(args)


def load_model(args):
    if args.tokenizer_name == "hash":
        tokenizer = HashTokenizer()
    else:
        from transformers import BertTokenizer
        tokenizer = BertTokenizer.from_pretrained(args.tokenizer_name or "bert-base-uncased", do_lower_case=True)
    model = BiBertEncoder(args)
    if args.model_name_or_path and os.path.exists(args.model_name_or_path):
        saved_state = load_states_from_checkpoint(args.model_name_or_path)
        model.load_state_dict(saved_state.model_dict, strict=False)
    teacher_model = get_bert_reader_components(args)
    if args.teacher_model_path and os.path.exists(args.teacher_model_path):
        saved_state = load_states_from_checkpoint(args.teacher_model_path)
        teacher_model.load_state_dict(saved_state.model_dict, strict=False)
    return tokenizer, model, teacher_model


def main(argv=None):
    args = get_arguments(argv)
    set_env(args)
    tokenizer, model, teacher_model = load_model(args)
    if args.output_dir and is_first_worker():
        os.makedirs(args.output_dir, exist_ok=True)
    if args.local_rank != -1:
        dist.barrier()
    global_step = train(args, model, teacher_model, tokenizer, args.global_step)
    logger.info(" global_step = %s", global_step)
    return global_step


if __name__ == "__main__":
    main()
