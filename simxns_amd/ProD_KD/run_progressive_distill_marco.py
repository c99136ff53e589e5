"""The cross-encoder distillation step of PROD/ProD_KD/run_progressive_distill_marco.py:288-333 (teacher_type ==
"cross_encoder") on the MI355X engine: student bi-encoder forward, frozen cross-encoder teacher forward, optional frozen
student copy (LwF), CrossBERTKDLoss, backward, clip + AdamW + schedule.  The script's data pipeline, logging and
checkpoint cadence are host-side and not rebuilt here; this is the step body a train loop calls."""
import torch

from .model.models import CrossBERTKDLoss


def cross_encoder_distill_step(args, model, teacher_model, inputs_retriever, inputs_reranker, student_copy=None,
                               optimizer=None, scheduler=None, world_size=1, last_micro_step=True):
    """-> (loss, is_correct).  ``inputs_retriever``: query_ids, attention_mask_q, input_ids_a, attention_mask_a;
    ``inputs_reranker``: input_ids [B,1+N,L], attention_mask.  With ``optimizer`` and ``last_micro_step`` the step is completed
    (clip_grad_norm_(args.max_grad_norm) -> optimizer.step -> scheduler.step -> zero_grad, :321-329); with
    gradient_accumulation_steps > 1 pass ``last_micro_step=False`` for all but the last micro-step: the optimiser is then
    neither stepped nor allowed to all-reduce the still-accumulating gradient buffers (FusedAdamW.armed)."""
    teacher_model.eval()
    if optimizer is not None:
        optimizer.armed = bool(last_micro_step)
    local_q_vector, local_ctx_vectors = model(**inputs_retriever)
    with torch.no_grad():
        binary_logits, relevance_logits, _ = teacher_model(**inputs_reranker)
    if getattr(args, "open_LwF", False):
        if student_copy is None:
            raise ValueError("--open_LwF needs the frozen student copy")
        student_copy.eval()
        with torch.no_grad():      # the copy is not in the optimiser (:196-205): its outputs are constants to the student
            ori_q_vector, ori_ctx_vectors = student_copy(**inputs_retriever)
        loss, is_correct = CrossBERTKDLoss().calc(args, local_q_vector, local_ctx_vectors, relevance_logits, LwF=True,
                                                  ori_q_vector=ori_q_vector, ori_ctx_vectors=ori_ctx_vectors)
    else:
        loss, is_correct = CrossBERTKDLoss().calc(args, local_q_vector, local_ctx_vectors, relevance_logits)
    loss = loss / getattr(args, "gradient_accumulation_steps", 1)
    loss.backward()
    if optimizer is not None and last_micro_step:
        optimizer.step(max_grad_norm=getattr(args, "max_grad_norm", 0.0), world_size=world_size)
        if scheduler is not None:
            scheduler.step()
    return loss, is_correct
