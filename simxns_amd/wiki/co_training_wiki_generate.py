"""Generate job of the NQ / TriviaQA AR2+SimANS iteration on the MI355X engine (SimANS/wiki/co_training_wiki_generate.py,
the second command of every train_NQ_AR2.sh / train_TQ_AR2.sh round): load output_dir/checkpoint-<global_step> (:259-266),
embed ``--passage_path`` and the train / dev / test questions, mine top-100 and write the round's files into ``--ann_dir``
(:268-292) -- ``train_ce_<step>.json`` is what the next train job samples its SimANS negatives from.  Flags as the
reference's (:30-180; the train job's parser covers them)."""
import logging
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simxns_amd.co_training import co_training_marco_train as M                               # noqa: E402
from simxns_amd.model.models import BiBertEncoder                                             # noqa: E402
from simxns_amd.utils.MARCO_until_new import HashTokenizer                                    # noqa: E402
from simxns_amd.utils.dpr_utils import get_model_obj, load_states_from_checkpoint             # noqa: E402
from simxns_amd.utils.util import is_first_worker                                             # noqa: E402
from simxns_amd.wiki.co_training_generate_new_train_wiki import RenewTools                    # noqa: E402

logger = logging.getLogger("__main__")


def load_model(args):
    if args.tokenizer_name == "hash":
        tokenizer = HashTokenizer()
    else:
        from transformers import BertTokenizer
        tokenizer = BertTokenizer.from_pretrained(args.tokenizer_name or "bert-base-uncased", do_lower_case=True)
    model = BiBertEncoder(args)
    if args.model_name_or_path and os.path.exists(args.model_name_or_path):
        model.load_state_dict(load_states_from_checkpoint(args.model_name_or_path).model_dict, strict=False)
    return tokenizer, model.to(args.device)


@torch.no_grad()
def get_new_dataset(args, model, global_step, renew_tools):
    if global_step != 0:
        path = os.path.join(args.output_dir, 'checkpoint-' + str(global_step))
        get_model_obj(model).load_state_dict(load_states_from_checkpoint(path).model_dict)
        logger.info(" model_path = %s", path)
    model.eval()
    group = dist.group.WORLD if args.world_size > 1 else None
    index = renew_tools.get_new_faiss_index(model, args.device)
    for mode, qa, gold in (("train", args.train_qa_path, args.origin_data_dir), ("dev", args.dev_qa_path, args.origin_data_dir_dev),
                           ("test", args.test_qa_path, args.origin_data_dir_dev)):
        if not qa:
            continue
        q, a, emb = renew_tools.get_question_embedding(model, args.device, qa)
        renew_tools.get_question_topk(q, a, emb, gold, index, mode=mode, step_num=global_step, group=group)


def main(argv=None):
    args = M.get_arguments(argv)
    M.set_env(args)
    if args.output_dir and is_first_worker():
        os.makedirs(args.output_dir, exist_ok=True)
    tokenizer, model = load_model(args)
    renew_tools = RenewTools(passages_path=args.passage_path, tokenizer=tokenizer, output_dir=args.ann_dir,
                             temp_dir=os.path.join(args.ann_dir, 'temp'), max_seq_length=args.max_seq_length, rank=args.rank,
                             world=args.world_size)
    if args.world_size > 1:
        dist.barrier()
    if args.global_step <= args.max_steps:
        get_new_dataset(args, model, args.global_step, renew_tools)
    logger.info(" global_step = %s", args.global_step)
    if args.world_size > 1:
        dist.barrier()
    return args.global_step


if __name__ == "__main__":
    main()
