"""Knowledge-distillation criterion (interface of the reference's ``quant/utils/kd_criterion.py:11-52``)."""

import torch
import torch.nn.functional as F


def kd_criterion(output_student: torch.Tensor, output_teacher: torch.Tensor, target: torch.Tensor, temperature: float,
                 freeze_teacher: bool = True, teacher_correction: bool = True) -> torch.Tensor:
    """Mean over the batch of T^2 * KL(softmax(teacher / T) || softmax(student / T)) per sample.

    ``teacher_correction`` swaps in the plain cross entropy for samples selected by the reference's mask
    ``pred_teacher.eq(pred_teacher)`` (kd_criterion.py:45) -- a comparison of the teacher's prediction with ITSELF, true
    everywhere, so the correction never fires there; kept bit for bit so that losses match the reference's."""
    teacher = output_teacher.detach() if freeze_teacher else output_teacher
    t2 = temperature * temperature
    per_class = F.kl_div(F.log_softmax(output_student / temperature, dim=1), F.softmax(teacher / temperature, dim=1),
                         reduction='none') * t2
    kd = per_class.sum(dim=1)
    if not teacher_correction:
        return kd.mean()
    pred = teacher.argmax(dim=1)
    keep = pred.eq(pred)
    ce = F.cross_entropy(output_student, target, reduction='none')
    return (keep * kd + ~keep * ce).mean()
