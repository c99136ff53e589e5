"""MS-MARCO Document train job (SimANS/Doc_training/co_training_doc_train.py, launched by train_MS_Doc_AR2.sh): the
MS-Pas loop with ONE shared RobertaDot student (query / document embeddings through the same RoBERTa + Linear +
LayerNorm head, :203-208, 640), a RoBERTa cross-encoder teacher (HFRobertaEncoder + Reranker, :76-84) and Doc_v2Dataset
(Gaussian SimANS weights, q128 / d512, pad id 1).  Same flags as the MS-Pas job plus --a / --b."""
import logging
import os
import sys

import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simxns_amd.co_training import co_training_marco_train as T                      # noqa: E402
from simxns_amd.engine import BertConfigLite                                          # noqa: E402
from simxns_amd.model.models import HFBertEncoder, Reranker, RobertaDot                # noqa: E402
from simxns_amd.utils.MARCO_until_Doc import Doc_v2Dataset, RobertaHashTokenizer       # noqa: E402
from simxns_amd.utils.dpr_utils import load_states_from_checkpoint                    # noqa: E402
from simxns_amd.utils.util import is_first_worker                                     # noqa: E402

logger = logging.getLogger(__name__)


def _roberta_cfg(path):
    cfg = BertConfigLite.from_pretrained(path)
    if cfg.position_offset == 0:                       # config.json without model_type: still a RoBERTa here
        cfg.pad_token_id, cfg.position_offset = 1, 2
    return cfg


def load_model(args):
    if args.tokenizer_name == "hash":
        tokenizer = RobertaHashTokenizer()
    else:
        from transformers import RobertaTokenizer
        tokenizer = RobertaTokenizer.from_pretrained(args.tokenizer_name or "roberta-base")
    dtype = os.environ.get("SIMX_DTYPE") or ("fp16" if args.fp16 else "fp32")      # as HFBertEncoder.init_encoder
    model = RobertaDot(_roberta_cfg(args.model_type), compute_dtype=dtype)
    if args.model_name_or_path and os.path.exists(args.model_name_or_path):
        model.load_state_dict(load_states_from_checkpoint(args.model_name_or_path).model_dict, strict=False)
    tcfg = _roberta_cfg(args.teacher_model_type)
    tcfg.add_pooling_layer = False
    teacher_model = Reranker(HFBertEncoder(tcfg, compute_dtype=dtype), tcfg.hidden_size)     # HFRobertaEncoder role
    if args.teacher_model_path and os.path.exists(args.teacher_model_path):
        teacher_model.load_state_dict(load_states_from_checkpoint(args.teacher_model_path).model_dict, strict=False)
    return tokenizer, model, teacher_model


def _dot_encode(model, q_ids, q_mask, c_ids, c_mask):
    return model(q_ids, q_mask, True), model(c_ids, c_mask, False)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a, b = 0.5, 0.0                                     # --a / --b are not flags of the MS-Pas parser
    for name in ("--a", "--b"):
        if name in argv:
            i = argv.index(name)
            v = float(argv[i + 1])
            del argv[i:i + 2]
            a, b = (v, b) if name == "--a" else (a, v)
    args = T.get_arguments(argv)
    T.set_env(args)
    tokenizer, model, teacher_model = load_model(args)
    if args.output_dir and is_first_worker():
        os.makedirs(args.output_dir, exist_ok=True)
    if args.local_rank != -1:
        dist.barrier()
    global_step = T.train(args, model, teacher_model, tokenizer, args.global_step, dataset_cls=Doc_v2Dataset,
                          dataset_kwargs=dict(a=a, b=b), encode_pair=_dot_encode)
    logger.info(" global_step = %s", global_step)
    return global_step


if __name__ == "__main__":
    main()
