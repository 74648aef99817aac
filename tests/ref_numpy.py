"""Second, independent reading of the reference's MultiBoxTarget / MultiBoxDetection
semantics in vectorised numpy (small cases only).  Written from
operator/multibox_target.cc:73-284 and operator/multibox_detection.cc:54-169
without looking at oracle/multibox_oracle.c's structure, so that the two
restatements check each other."""
import numpy as np

f32 = np.float32


_T = None


def expf(x):
    """glibc expf (e_expf.c) modelled in float64: 32-entry 2^(i/32) table times a cubic."""
    global _T
    if _T is None:
        _T = (np.exp2(np.arange(32, dtype=np.longdouble) / 32).astype(np.float64).view(np.uint64)
              - (np.arange(32, dtype=np.uint64) << np.uint64(47)))
    xd = np.asarray(x, np.float32).astype(np.float64)
    z = (float.fromhex('0x1.71547652b82fep+0') * 32) * xd
    kd = z + float.fromhex('0x1.8p+52')
    ki = kd.view(np.uint64) if kd.ndim else np.array(kd).view(np.uint64)
    kd = kd - float.fromhex('0x1.8p+52')
    r = z - kd
    t = _T[(ki & np.uint64(31)).astype(np.int64)] + (ki << np.uint64(47))
    s = t.view(np.float64)
    c0, c1, c2 = (float.fromhex('0x1.c6af84b912394p-5') / 32 / 32 / 32,
                  float.fromhex('0x1.ebfce50fac4f3p-3') / 32 / 32,
                  float.fromhex('0x1.62e42ff0c52d6p-1') / 32)
    y = ((c0 * r + c1) * (r * r) + (c2 * r + 1.0)) * s
    return y.astype(f32)


def logf(x):
    return np.log(np.asarray(x, np.float64)).astype(f32)


def iou_matrix(anchors, gts):
    a = anchors[:, None, :].astype(f32)
    g = gts[None, :, :].astype(f32)
    iw = np.maximum(f32(0), np.minimum(a[..., 2], g[..., 2]) - np.maximum(a[..., 0], g[..., 0]))
    ih = np.maximum(f32(0), np.minimum(a[..., 3], g[..., 3]) - np.maximum(a[..., 1], g[..., 1]))
    inter = (iw * ih).astype(f32)
    uni = ((a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
           + (g[..., 2] - g[..., 0]) * (g[..., 3] - g[..., 1]) - inter).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where(uni == 0, f32(0), inter / uni)
    return out.astype(f32)


def target(anchor, label, cls_pred, overlap_threshold=0.5, ignore_label=-1.0,
           negative_mining_ratio=-1.0, negative_mining_thresh=0.5,
           variances=(0.1, 0.1, 0.2, 0.2)):
    anchors = anchor.reshape(-1, 4).astype(f32)
    B, L, _ = label.shape
    A = anchors.shape[0]
    loc_t = np.zeros((B, A, 5), f32)
    loc_m = np.zeros((B, A, 5), f32)
    cls_t = np.full((B, A), ignore_label, f32)
    for b in range(B):
        cls_col = label[b, :, 0]
        stop = np.nonzero(cls_col == -1)[0]
        G = int(stop[0]) if len(stop) else L
        if G == 0:
            continue
        gts = label[b, :G]
        ov = iou_matrix(anchors, gts[:, 1:5])        # (A, G)
        flag = -np.ones(A, np.int64)
        match = -np.ones(A, np.int64)
        best_iou = -np.ones(A, f32)
        gt_done = np.zeros(G, bool)
        work = ov.copy()
        while not gt_done.all():
            masked = np.where((flag == 1)[:, None] | gt_done[None, :], f32(-1), work)
            flat = int(np.argmax(masked))            # first maximum in (anchor, gt) order
            j, k = divmod(flat, G)
            if not masked[j, k] > f32(1e-6):
                break
            flag[j] = 1; match[j] = k; best_iou[j] = masked[j, k]; gt_done[k] = True
        rest = flag != 1
        row_arg = np.argmax(ov, axis=1)
        row_max = ov[np.arange(A), row_arg]
        match[rest] = row_arg[rest]
        best_iou[rest] = row_max[rest]
        if overlap_threshold > 0:
            pos = rest & (row_max > f32(overlap_threshold))
            flag[pos] = 1
        npos = int((flag == 1).sum())
        if negative_mining_ratio > 0:
            nneg = int(f32(npos) * f32(negative_mining_ratio))
            nneg = min(nneg, A - npos)
            if nneg > 0:
                cand = np.nonzero((flag == -1) & (best_iou < f32(negative_mining_thresh)))[0]
                p = cls_pred[b][:, cand].astype(f32)             # (C, n)
                mx = p.max(axis=0)
                e = expf((p - mx).astype(f32))
                s = np.zeros_like(mx)
                for c in range(e.shape[0]):                      # sequential float sum
                    s = (s + e[c]).astype(f32)
                prob = (e[0] / s).astype(f32)
                order = np.argsort(prob, kind="stable")          # ascending P(bg), ties by index
                flag[cand[order[:nneg]]] = 0
        else:
            flag[flag != 1] = 0
        vx, vy, vw, vh = [f32(v) for v in variances]
        for j in np.nonzero(flag == 1)[0]:
            g = gts[match[j]]
            al, at, ar, ab = anchors[j]
            aw, ah = f32(ar - al), f32(ab - at)
            ax, ay = f32((al + ar) * 0.5), f32((at + ab) * 0.5)
            gw, gh = f32(g[3] - g[1]), f32(g[4] - g[2])
            gx, gy = f32((g[1] + g[3]) * 0.5), f32((g[2] + g[4]) * 0.5)
            loc_t[b, j] = [f32(f32(gx - ax) / aw) / vx, f32(f32(gy - ay) / ah) / vy,
                           f32(logf(f32(gw / aw))) / vw, f32(logf(f32(gh / ah))) / vh,
                           f32(np.float64(g[5]) / 0.1)]
            loc_m[b, j] = 1
            cls_t[b, j] = g[0] + 1
        cls_t[b, flag == 0] = 0
    return [loc_t.reshape(B, A * 5), loc_m.reshape(B, A * 5), cls_t]


def detection(cls_prob, loc_pred, anchor, clip=True, threshold=0.01, nms_threshold=0.5,
              force_suppress=False, variances=(0.1, 0.1, 0.2, 0.2), nms_topk=-1):
    anchors = anchor.reshape(-1, 4).astype(f32)
    B, C, A = cls_prob.shape
    out = -np.ones((B, A, 7), f32)
    vx, vy, vw, vh = [f32(v) for v in variances]
    for b in range(B):
        fg = cls_prob[b, 1:, :]
        if C > 1:
            cid = np.argmax(fg, axis=0)               # first max == strict '>' scan
            score = fg[cid, np.arange(A)]
            keep = score >= f32(threshold)
            keep &= score > f32(-1)
        else:
            cid = np.zeros(A, np.int64); score = -np.ones(A, f32); keep = np.zeros(A, bool)
        idx = np.nonzero(keep)[0]
        V = len(idx)
        a = anchors[idx]
        p = loc_pred[b].reshape(A, 5)[idx].astype(f32)
        aw, ah = (a[:, 2] - a[:, 0]).astype(f32), (a[:, 3] - a[:, 1]).astype(f32)
        ax, ay = ((a[:, 0] + a[:, 2]) / f32(2)).astype(f32), ((a[:, 1] + a[:, 3]) / f32(2)).astype(f32)
        ox = ((p[:, 0] * vx).astype(f32) * aw + ax).astype(f32)
        oy = ((p[:, 1] * vy).astype(f32) * ah + ay).astype(f32)
        ow = ((expf((p[:, 2] * vw).astype(f32)) * aw).astype(f32) / f32(2)).astype(f32)
        oh = ((expf((p[:, 3] * vh).astype(f32)) * ah).astype(f32) / f32(2)).astype(f32)
        oz = (p[:, 4].astype(np.float64) * 0.1).astype(f32)
        rows = np.stack([cid[idx].astype(f32), score[idx], ox - ow, oy - oh, ox + ow, oy + oh, oz],
                        axis=1).astype(f32)
        if clip:
            rows[:, 2:] = np.clip(rows[:, 2:], 0, 1)
        out[b, :V] = rows
        if V < 1 or nms_threshold <= 0 or nms_threshold > 1:
            continue
        order = np.argsort(-rows[:, 1], kind="stable")
        nkeep = V if not (0 < nms_topk < V) else nms_topk
        cur = rows.copy()
        cur[:nkeep] = rows[order[:nkeep]]
        for i in range(V):
            if cur[i, 0] < 0:
                continue
            later = np.arange(i + 1, V)
            if len(later) == 0:
                break
            live = cur[later, 0] >= 0
            same = live & (force_suppress | (cur[later, 0] == cur[i, 0]))
            j = later[same]
            if len(j) == 0:
                continue
            bi, bj = cur[i, 2:6], cur[j, 2:6]
            w = np.maximum(f32(0), np.minimum(bi[2], bj[:, 2]) - np.maximum(bi[0], bj[:, 0]))
            h = np.maximum(f32(0), np.minimum(bi[3], bj[:, 3]) - np.maximum(bi[1], bj[:, 1]))
            inter = (w * h).astype(f32)
            u = ((bi[2] - bi[0]) * (bi[3] - bi[1]) + (bj[:, 2] - bj[:, 0]) * (bj[:, 3] - bj[:, 1])
                 - inter).astype(f32)
            with np.errstate(divide="ignore", invalid="ignore"):
                iou = np.where(u <= 0, f32(0), inter / u).astype(f32)
            cur[j[iou >= f32(nms_threshold)], 0] = -1
        out[b, :V] = cur
    return out
