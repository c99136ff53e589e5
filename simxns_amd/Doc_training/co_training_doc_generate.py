"""Generate job of the MS-MARCO Document iteration on the MI355X engine (SimANS/Doc_training/co_training_doc_generate.py,
second command of every train_MS_Doc_AR2.sh round): load output_dir/checkpoint-<global_step> into the shared RobertaDot
(:259-266), embed ``<passage_path>/msmarco-docs.tsv`` and the train / dev queries, mine and write ``train_ce_<step>.tsv`` /
``dev_ce_<step>.tsv`` into ``--ann_dir`` (:267-288; the qrels files sit beside the query files)."""
import logging
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simxns_amd.Doc_training import co_training_doc_train as T                                 # noqa: E402
from simxns_amd.Doc_training.co_training_generate_new_train import RenewTools                  # noqa: E402
from simxns_amd.co_training import co_training_marco_train as M                                # noqa: E402
from simxns_amd.utils.dpr_utils import get_model_obj, load_states_from_checkpoint              # noqa: E402
from simxns_amd.utils.util import is_first_worker                                              # noqa: E402

logger = logging.getLogger("__main__")


@torch.no_grad()
def get_new_dataset(args, model, global_step, renew_tools):
    path = os.path.join(args.output_dir, 'checkpoint-' + str(global_step))
    get_model_obj(model).load_state_dict(load_states_from_checkpoint(path).model_dict, strict=False)
    logger.info(" model_path = %s", path)
    model.eval()
    group = dist.group.WORLD if args.world_size > 1 else None
    index = renew_tools.get_new_faiss_index(model, args.device)
    out = {}
    for mode, qa in (("train", args.train_qa_path), ("dev", args.dev_qa_path)):
        if not qa:
            continue
        gold = os.path.join(os.path.abspath(os.path.dirname(qa)), 'msmarco-doc%s-qrels.tsv' % mode)
        q, qids, emb = renew_tools.get_question_embedding(model, args.device, qa)
        out[mode] = renew_tools.get_question_topk(q, qids, emb, gold, index, mode=mode, step_num=global_step, group=group)
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    max_doc_character = 10000
    if "--max_doc_character" in argv:
        i = argv.index("--max_doc_character")
        max_doc_character = int(argv[i + 1])
        del argv[i:i + 2]
    args = M.get_arguments(argv)
    M.set_env(args)
    if not args.teacher_model_type:
        args.teacher_model_type = args.model_type        # (the generate job has no teacher; load_model wants a config path)
    if args.output_dir and is_first_worker():
        os.makedirs(args.output_dir, exist_ok=True)
    tokenizer, model, _ = T.load_model(args)
    model.to(args.device)
    renew_tools = RenewTools(os.path.join(args.passage_path, 'msmarco-docs.tsv'), tokenizer, args.ann_dir,
                             temp_dir=os.path.join(args.ann_dir, 'temp'), max_doc_character=max_doc_character, rank=args.rank,
                             world=args.world_size)
    if args.world_size > 1:
        dist.barrier()
    out = get_new_dataset(args, model, args.global_step, renew_tools) if args.global_step <= args.max_steps else {}
    logger.info(" global_step = %s", args.global_step)
    if args.world_size > 1:
        dist.barrier()
    return out


if __name__ == "__main__":
    main()
