"""Part-segmentation evaluation: the metrics of the reference's testing.evaluation
(testing.py:49-249) computed batched on the device instead of per-shape / per-label numpy loops.

    ev = SegmentationEvaluator()                     # 50 parts, 16 categories (testing.py:30-39)
    ev.update(seg_pred [B,N,50] logits, target [B,N])
    ev.compute() -> {'accuracy', 'class_avg_accuracy', 'class_avg_iou', 'instance_avg_iou', 'category_iou': {...}}

    evaluation(model, loader, ...)                   # the loop of testing.py:110-137 around a model of this package

Rules kept from the reference: the category of a shape is that of its first target label (:142); the prediction
is the arg-max over that category's parts only (:143-144); a part absent from both prediction and target counts
as IoU 1 (:203-204); a shape's IoU is the mean over its category's parts; class_avg_iou averages the per-category
means, instance_avg_iou averages all shapes (:226-240)."""
import numpy as np
import torch

seg_classes = {'Airplane': [0, 1, 2, 3], 'Bag': [4, 5], 'Cap': [6, 7], 'Car': [8, 9, 10, 11],
               'Chair': [12, 13, 14, 15], 'Earphone': [16, 17, 18], 'Guitar': [19, 20, 21], 'Knife': [22, 23],
               'Lamp': [24, 25, 26, 27], 'Laptop': [28, 29], 'Motorbike': [30, 31, 32, 33, 34, 35], 'Mug': [36, 37],
               'Pistol': [38, 39, 40], 'Rocket': [41, 42, 43], 'Skateboard': [44, 45, 46], 'Table': [47, 48, 49]}
classes = ['Airplane', 'Bag', 'Cap', 'Car', 'Chair', 'Earphone', 'Guitar', 'Knife', 'Lamp', 'Laptop', 'Motorbike',
           'Mug', 'Pistol', 'Rocket', 'Skateboard', 'Table']


def to_categorical(y, num_classes):
    """1-hot encodes a tensor (testing.py:42-47), on y's device."""
    return torch.eye(num_classes, device=y.device)[y]


class SegmentationEvaluator:
    def __init__(self, num_part=50, categories=None):
        self.categories = categories if categories is not None else seg_classes
        self.names = list(self.categories.keys())
        self.num_part = num_part
        cat_of_label = torch.full((num_part,), -1, dtype=torch.long)
        for ci, name in enumerate(self.names):
            cat_of_label[self.categories[name]] = ci
        self.cat_of_label = cat_of_label
        self.reset()

    def reset(self):
        self.correct = 0
        self.seen = 0
        self.seen_class = torch.zeros(self.num_part, dtype=torch.long)
        self.correct_class = torch.zeros(self.num_part, dtype=torch.long)
        self.shape_iou, self.shape_cat = [], []

    @torch.no_grad()
    def predict(self, seg_pred, target):
        """arg-max restricted to the parts of each shape's category (testing.py:141-144)."""
        col = self.cat_of_label.to(seg_pred.device)
        cat = col[target[:, 0]]                                            # [B]
        allowed = col.view(1, 1, -1) == cat.view(-1, 1, 1)                 # [B,1,P]
        return seg_pred.masked_fill(~allowed, float('-inf')).argmax(dim=-1), cat

    @torch.no_grad()
    def update(self, seg_pred, target):
        B, N, P = seg_pred.shape
        target = target.long()
        pred, cat = self.predict(seg_pred, target)
        hit = pred == target
        self.correct += int(hit.sum())
        self.seen += B * N
        self.seen_class += torch.bincount(target.reshape(-1), minlength=P).cpu()
        self.correct_class += torch.bincount(target[hit], minlength=P).cpu()
        oh_p = torch.nn.functional.one_hot(pred, P)
        oh_t = torch.nn.functional.one_hot(target, P)
        inter = (oh_p & oh_t).sum(dim=1).double()                          # [B,P]
        union = (oh_p | oh_t).sum(dim=1).double()
        part_iou = torch.where(union == 0, torch.ones_like(union), inter / union.clamp(min=1.0))
        in_cat = (self.cat_of_label.to(seg_pred.device).view(1, -1) == cat.view(-1, 1)).double()
        self.shape_iou.append(((part_iou * in_cat).sum(dim=1) / in_cat.sum(dim=1)).cpu())
        self.shape_cat.append(cat.cpu())
        return pred

    def compute(self):
        iou = torch.cat(self.shape_iou) if self.shape_iou else torch.zeros(0, dtype=torch.float64)
        cat = torch.cat(self.shape_cat) if self.shape_cat else torch.zeros(0, dtype=torch.long)
        per_cat = {}
        for ci, name in enumerate(self.names):
            sel = iou[cat == ci]
            per_cat[name] = float(sel.mean()) if sel.numel() else float('nan')
        seen = self.seen_class.double()
        with np.errstate(invalid='ignore', divide='ignore'):
            class_acc = np.mean((self.correct_class.double() / seen).numpy())   # nan if a part never occurs (:229-230)
        present = [v for v in per_cat.values() if not np.isnan(v)]
        return {'accuracy': self.correct / float(max(self.seen, 1)),
                'class_avg_accuracy': float(class_acc),
                # mean over categories (:226,:234); categories without shapes are left out, which equals the
                # reference whenever every category occurs (it would produce nan otherwise)
                'class_avg_iou': float(np.mean(present)) if present else float('nan'),
                'instance_avg_iou': float(iou.mean()) if iou.numel() else float('nan'),
                'category_iou': per_cat}


@torch.no_grad()
def evaluation(classifier, loader, num_classes=16, num_part=50, category=True, device=None, metrics=None, epoch=0,
               **forward_kwargs):
    """testing.py:110-249 for a model of this package: loader yields (points [B,N,C], label [B,1], target [B,N]).
    Returns the test metrics; if `metrics` (the running best, :241-247) is given it is updated in place."""
    device = device if device is not None else next(classifier.parameters()).device
    ev = SegmentationEvaluator(num_part)
    was_training = classifier.training
    classifier.eval()
    chamfer = []
    for points, label, target in loader:
        points = torch.as_tensor(points).float().to(device).transpose(2, 1).contiguous()
        label = torch.as_tensor(label).long().to(device)
        target = torch.as_tensor(target).long().to(device)
        if category:
            category_label = to_categorical(label, num_classes).contiguous()
        else:
            category_label = torch.zeros(label.shape[0], 1, num_classes, device=device)
        out = classifier(points, category_label, include_convex_loss=False, evaluation=True, **forward_kwargs)
        seg_pred = out[0]
        if len(out) >= 5 and torch.is_tensor(out[4]):
            chamfer.append(float(out[4].float().mean()))
        ev.update(seg_pred, target)
    classifier.train(was_training)
    test_metrics = ev.compute()
    test_metrics['chamfer_loss'] = float(np.mean(chamfer)) if chamfer else 0.0
    if metrics:
        if metrics.get('best_class_avg_miou', -1.0) <= test_metrics['class_avg_iou']:
            metrics.update(best_chamfer_loss=test_metrics['chamfer_loss'], best_epoch=epoch + 1,
                           best_acc=test_metrics['accuracy'], best_class_avg_miou=test_metrics['class_avg_iou'],
                           best_instance_avg_miou=test_metrics['instance_avg_iou'])
    return test_metrics
