"""Mean intersection-over-union accumulator (reference: src/myrtle_vision/utils/miou.py), float64 totals."""
import torch


def intersect_and_union(pred_label, label, num_classes):
    pred_label, label = pred_label.reshape(-1).long(), label.reshape(-1).long()
    inter = pred_label[pred_label == label]
    area_intersect = torch.bincount(inter, minlength=num_classes)[:num_classes].double()
    area_pred = torch.bincount(pred_label.clamp(0, num_classes - 1), minlength=num_classes)[:num_classes].double()
    area_label = torch.bincount(label.clamp(0, num_classes - 1), minlength=num_classes)[:num_classes].double()
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
