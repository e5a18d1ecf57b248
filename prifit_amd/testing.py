"""Part-segmentation evaluation: the metrics of the reference's testing.evaluation
(testing.py:49-249) computed batched on the device instead of per-shape / per-label numpy loops.

    ev = SegmentationEvaluator()                     # 50 parts, 16 categories (testing.py:30-39)
    ev.update(seg_pred [B,N,50] logits, target [B,N])
    ev.compute() -> {'accuracy', 'class_avg_accuracy', 'class_avg_iou', 'instance_avg_iou', 'category_iou': {...}}

    evaluation(args, epoch, classifier, metrics)     # testing.py:49, the reference's signature (train_partseg_shapenet.py:487)
    evaluate_loader(model, loader, ...)              # the loop of testing.py:110-137 around a model and any iterable of batches

Rules kept from the reference: the category of a shape is that of its first target label (:142); the prediction
is the arg-max over that category's parts only (:143-144); a part absent from both prediction and target counts
as IoU 1 (:203-204); a shape's IoU is the mean over its category's parts; class_avg_iou averages the per-category
means, instance_avg_iou averages all shapes (:226-240)."""
import importlib
import os

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


DATA_ROOT = 'ShapeSelfSup/dataset/shapenetcore_partanno_segmentation_benchmark_v0_normal'   # testing.py:72, relative to the cwd


def evaluation(args, epoch=0, classifier=None, metrics={}, loader=None):   # noqa: B006 (the reference's signature)
    """testing.py:49-249 under the reference's signature: `evaluation(args, epoch, classifier, metrics)` as the trainer calls
    it after every epoch (train_partseg_shapenet.py:487), `evaluation(args)` as `python testing.py` does (:253-255).

    args: the namespace of args_parser.py -- eval_split, npoint, normal, batch_size, num_classes, num_parts, category, seed and,
    when `classifier` is None, model / pretrained_model (+ dgcnn_k, reconstruct) to build and load one (:96-110).  The test
    split is read from DATA_ROOT below the current directory (:72-77) unless `loader` is given.  The forward keywords of
    :139 (quantile, msc_iterations, max_num_clusters, alpha, beta, if_cuboid, embed, seed) are taken from `args` where present.
    Returns `metrics` -- the running best, updated in place when the class-average mIoU did not fall (:241-247) -- like the
    reference; the metrics of THIS evaluation are left in `evaluation.last`."""
    from . import data as D
    if getattr(args, "gpu", None) is not None:
        os.environ["CUDA_VISIBLE_DEVICES"] = args.gpu              # :55-56
    if loader is None:
        ds = D.PartNormalDataset(root=DATA_ROOT, npoints=args.npoint, split=args.eval_split, normal_channel=args.normal)
        loader = torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=False, num_workers=0)
        print("The number of test data is: %d" % len(ds))
    if classifier is None:
        if getattr(args, "pretrained_model", None) is None:
            raise ValueError("evaluation(args): no classifier given and args.pretrained_model is not set")
        MODEL = importlib.import_module(args.model)
        if 'dgcnn' in args.model:
            classifier = MODEL.get_model(args.num_parts, normal_channel=args.normal, k=args.dgcnn_k).cuda()
        else:
            classifier = MODEL.get_model(args.num_parts, normal_channel=args.normal).cuda()
        ckpt = torch.load(args.pretrained_model, map_location="cuda")
        state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in ckpt['model_state_dict'].items()}
        classifier.load_state_dict(state)                        # (upstream wraps in DataParallel first: 'module.' keys)
    fwd = {k: getattr(args, k) for k in ("quantile", "msc_iterations", "max_num_clusters", "alpha", "beta", "if_cuboid",
                                          "embed", "seed") if hasattr(args, k)}
    test_metrics = evaluate_loader(classifier, loader, num_classes=args.num_classes, num_part=args.num_parts,
                                   category=bool(args.category), metrics=metrics if metrics != {} else None, epoch=epoch, **fwd)
    evaluation.last = test_metrics
    if metrics != {}:
        print('Best test Accuracy: {:6f}, Best Epoch: {},  Best Class avg mIOU: {:6f}, Best Instance avg mIOU: {:6f}, Best Loss: {:6f}'
              .format(metrics['best_acc'], metrics['best_epoch'], metrics['best_class_avg_miou'],
                      metrics['best_instance_avg_miou'], metrics['best_chamfer_loss']))
    print('%sAccuracy: %f  Class avg mIOU: %f   Instance avg mIOU: %f, Loss: %f' % (
        ('Epoch %d test ' % (epoch + 1)) if metrics != {} else 'Test ', test_metrics['accuracy'], test_metrics['class_avg_iou'],
        test_metrics['instance_avg_iou'], test_metrics['chamfer_loss']))
    return metrics


@torch.no_grad()
def evaluate_loader(classifier, loader, num_classes=16, num_part=50, category=True, device=None, metrics=None, epoch=0,
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
        # (the reference passes the one-hot label whatever args.category says, :139: its `category_label` of :131-134 is
        # computed and never used; `category` is kept in the signature for the same reason)
        out = classifier(points, to_categorical(label, num_classes), include_convex_loss=False, evaluation=True, **forward_kwargs)
        seg_pred = out[0]
        if len(out) >= 5 and torch.is_tensor(out[4]):
            chamfer.append(float(out[4].float().mean()))
        ev.update(seg_pred, target)
    classifier.train(was_training)
    test_metrics = ev.compute()
    test_metrics['chamfer_loss'] = float(np.mean(chamfer)) if chamfer else 0.0
    if metrics:
        if metrics['best_class_avg_miou'] <= test_metrics['class_avg_iou']:
            metrics.update(best_chamfer_loss=test_metrics['chamfer_loss'], best_epoch=epoch + 1,
                           best_acc=test_metrics['accuracy'], best_class_avg_miou=test_metrics['class_avg_iou'],
                           best_instance_avg_miou=test_metrics['instance_avg_iou'])
    return test_metrics
