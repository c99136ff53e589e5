"""Runtime utilities of the path with the reference's names (SimANS/utils/dpr_utils.py): checkpoint state,
model unwrapping and the gather helper.  FAISS indexers, Eval_Tool and answer matching are out of scope
(evaluation side)."""
import collections
import logging

import torch

from ..parallel import all_gather_list  # noqa: F401  (same name / call signature as dpr_utils.py:166)

logger = logging.getLogger()

# dpr_utils.py:22-24 -- on-disk layout: torch.save(state._asdict(), path)
CheckpointState = collections.namedtuple("CheckpointState",
                                         ['model_dict', 'optimizer_dict', 'scheduler_dict', 'offset', 'epoch',
                                          'encoder_params'])


def get_model_obj(model):
    """dpr_utils.py:57-58"""
    return model.module if hasattr(model, 'module') else model


def load_states_from_checkpoint(model_file: str) -> CheckpointState:
    """dpr_utils.py:73-77"""
    logger.info('Reading saved model from %s', model_file)
    state_dict = torch.load(model_file, map_location='cpu', weights_only=False)
    logger.info('model_state_dict keys %s', state_dict.keys())
    return CheckpointState(**state_dict)


def save_checkpoint_state(path, model, optimizer, scheduler, offset=0, epoch=0, encoder_params=None):
    """co_training_marco_train.py:310-345 (_save_checkpoint / _save_teacher_checkpoint)."""
    model_to_save = get_model_obj(model)
    state = CheckpointState(model_to_save.state_dict(), optimizer.state_dict(), scheduler.state_dict(), offset, epoch,
                            encoder_params)
    torch.save(state._asdict(), path)
    return path
