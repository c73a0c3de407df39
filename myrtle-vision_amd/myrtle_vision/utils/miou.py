"""Mean intersection-over-union accumulator (reference: src/myrtle_vision/utils/miou.py), float64 totals."""
import torch


def _hist(v, num_classes):
    """torch.histc(v.float(), bins=C, min=0, max=C-1) of integer labels (reference miou.py:35-40): values outside
    [0, C-1] are DROPPED, not clamped into the end bins."""
    v = v[(v >= 0) & (v < num_classes)]
    return torch.bincount(v, minlength=num_classes)[:num_classes].double()


def intersect_and_union(pred_label, label, num_classes):
    pred_label, label = pred_label.reshape(-1).long(), label.reshape(-1).long()
    area_intersect = _hist(pred_label[pred_label == label], num_classes)
    area_pred = _hist(pred_label, num_classes)
    area_label = _hist(label, num_classes)
    return area_intersect, area_pred + area_label - area_intersect, area_pred, area_label


class MIoU:
    def __init__(self, num_classes, device):
        self.num_classes = num_classes
        self.total_area_intersect = torch.zeros(num_classes, dtype=torch.float64, device=device)
        self.total_area_union = torch.zeros(num_classes, dtype=torch.float64, device=device)

    def add_img(self, prediction_img, ground_truth_img):
        i, u, _, _ = intersect_and_union(prediction_img, ground_truth_img, self.num_classes)
        self.total_area_intersect += i.to(self.total_area_intersect.device)
        self.total_area_union += u.to(self.total_area_union.device)

    def get_per_class_iou(self):
        return self.total_area_intersect / self.total_area_union

    def get_miou(self):
        return torch.mean(self.get_per_class_iou()).item()
