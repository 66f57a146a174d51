"""TEST INFRASTRUCTURE - CPU oracle #1: PyTorch-eager restatement of the reference's
EM-Dirichlet / Hard EM-Dirichlet loop.

Not part of the product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this file; the product path (transductive-clip_amd/) never does and fails
loudly when its HIP library is missing.

What it is: a functional restatement of the algorithm that issues the SAME torch CPU ops in
the SAME order as the reference (same broadcast (N,Q,K,K) temporaries, same reductions, fp32),
so on one machine it reproduces the reference bit for bit - the special functions are torch's
own (calc_digamma, Sleef lgammaf/logf/expf) and the summation orders are torch's.  That is what
licenses timing it as "the reference's CPU path" in bench.py (kind "port").

Pinned against: tests/golden/*.npz, produced by running /root/reference itself in the build
container (tests/golden/make_golden.py); tests/test_oracle_golden.py demands bit-exact
alpha/u/v/criterions and identical MM iteration counts on every small fixture.

Reference lines followed (paths relative to the reference repo):
  zero-shot  src/methods/zero_shot/em_dirichlet.py:28-40 (logits), :132-143 (E-step),
             :145-151 (v), :153-155 (curvature), :157-177 (MM loop), :195-246 (outer loop)
  hard       src/methods/zero_shot/hard_em_dirichlet.py:255-258
  few-shot   src/methods/few_shot/em_dirichlet.py:28-39, :167-220;
             src/methods/few_shot/hard_em_dirichlet.py:228-244
"""
import time

import torch

EPS = 1e-15
MM_CHECK_EVERY = 50
MM_TOL = 1e-11


def one_hot_rows(labels, n_class):
    """(N,S) int64 -> (N,S,K) f32, as src/utils.py:18-24 builds it (rows of an identity)."""
    eye = torch.eye(n_class)
    return torch.stack([eye[row] for row in labels], 0)


def mm_solve(alpha, y_cst, iter_mm, pi2_6, lgamma_1):
    """Majorize-minimize fixed point for the Dirichlet parameters, whole batch at once.

    Returns (alpha_new, number of MM iterations executed).  The stop test couples every task
    of the batch: squared Frobenius norms over the entire (N,K,K) tensor, every 50 iterations.
    """
    beta = alpha.clone()
    executed = 0
    for l in range(iter_mm):
        psi1 = torch.polygamma(0, beta + 1)
        curv = torch.where(beta > 1e-11,
                           abs(2 * (lgamma_1 - torch.lgamma(beta + 1) + psi1 * beta) / beta ** 2),
                           pi2_6)
        b = psi1 - torch.polygamma(0, beta.sum(-1)).unsqueeze(-1) - curv * beta
        b = b - y_cst
        delta = b ** 2 + 4 * curv
        beta_next = (-b + torch.sqrt(delta)) / (2 * curv)
        executed += 1
        if l > 0 and l % MM_CHECK_EVERY == 0:
            crit = torch.norm(beta_next - beta) ** 2 / torch.norm(beta) ** 2
            if crit < MM_TOL:
                break
        beta = beta_next.clone()
    return beta_next.clone(), executed


def dirichlet_logits(alpha, log_samples):
    """log Dirichlet density of every sample under every class, (N,n,K)."""
    l1 = torch.lgamma(alpha.sum(-1)).unsqueeze(1)
    l2 = -torch.lgamma(alpha).sum(-1).unsqueeze(1)
    l3 = ((alpha.unsqueeze(1) - 1) * log_samples.unsqueeze(2)).sum(-1)
    return l1 + l2 + l3


def run(x_q, x_s=None, y_s=None, *, n_class, iters, iter_mm=1000, lambd, hard=False, trace=None):
    """Runs the loop on CPU.  x_q (N,Q,K) probability features; x_s/y_s given = few-shot.

    lambd is the reference's integer `int(K/5)*n_query` (zero-shot) or `int(K/k_eff)*n_query`.
    Returns dict(u, v, alpha, criterions (iters,), mm_iters list, seconds, seconds_mm / seconds_iter: host wall time of
    every outer iteration's MM loop and of the whole outer iteration - what bench.py's cpu_baseline extrapolates from).
    `trace`, if a dict, receives per-iteration 'argmax' and 'live' lists.
    """
    few = x_s is not None
    query = x_q.clone().float()
    n_task, n_query = query.shape[0], query.shape[1]
    pi2_6 = torch.polygamma(1, torch.Tensor([1])).float()
    lgamma_1 = torch.lgamma(torch.Tensor([1])).float()

    v = torch.zeros(n_task, n_class)
    u = query.clone()
    alpha = torch.ones((n_task, n_class, n_class))
    alpha_old = alpha.clone()
    if few:
        support = x_s.clone().float()
        ys_hot = one_hot_rows(y_s.long().view(n_task, -1), n_class)
        support.add_(EPS).log_()
        query.add_(EPS).log_()          # few-shot works on log-features from here on
        log_q = query
    criterions, mm_iters, seconds_mm, seconds_iter = [], [], [], []
    t0 = time.time()
    _mm_solve = mm_solve

    def mm_solve_timed(*a):
        t = time.time()
        r = _mm_solve(*a)
        seconds_mm.append(time.time() - t)
        return r
    for _ in range(iters):
        t_iter = time.time()
        if few:
            denom = (1 / (ys_hot.sum(dim=1) + u.sum(dim=1))).unsqueeze(-1)
            y_cst = denom * ((ys_hot.unsqueeze(-1) * support.unsqueeze(2)).sum(dim=1)
                             + (u.unsqueeze(-1) * log_q.unsqueeze(2)).sum(dim=1))
            alpha, n_mm = mm_solve_timed(alpha, y_cst, iter_mm, pi2_6, lgamma_1)
        else:
            sizes = u.sum(dim=1).unsqueeze(-1).float()
            live = sizes > EPS
            y_cst = ((u.unsqueeze(-1) * torch.log(query + EPS).unsqueeze(2)).sum(1)
                     / u.sum(1).clamp(min=EPS).unsqueeze(-1))
            y_cst = y_cst * live + (1 - 1 * live) * torch.ones_like(y_cst) * (-10)
            alpha, n_mm = mm_solve_timed(alpha, y_cst, iter_mm, pi2_6, lgamma_1)
            alpha = alpha * live + alpha_old * (1 - 1 * live)
            if trace is not None:
                trace.setdefault("live", []).append(live.squeeze(-1).clone())
        mm_iters.append(n_mm)
        # v uses the responsibilities from BEFORE this iteration's E-step
        v = torch.log(u.sum(1) / u.size(1) + EPS) + 1
        logits = dirichlet_logits(alpha, log_q if few else torch.log(query + EPS))
        u = (logits + lambd * v.unsqueeze(1) / n_query).softmax(2)
        if hard:
            labels = torch.argmax(u, dim=-1)
            u.zero_()
            u.scatter_(2, labels.unsqueeze(-1), 1.0)
        if trace is not None:
            trace.setdefault("argmax", []).append(u.argmax(2).clone())
        crit = ((alpha_old - alpha).norm(dim=(1, 2)) / alpha_old.norm(dim=(1, 2))).mean(0)
        alpha_old = alpha.clone()
        if few and hard:
            # few_shot/hard_em_dirichlet.py:233-244 evaluates the criterion a second time after
            # alpha_old was refreshed, so what it logs is identically 0.
            crit = ((alpha_old - alpha).norm(dim=(1, 2)) / alpha_old.norm(dim=(1, 2))).mean(0)
        criterions.append(crit)
        seconds_iter.append(time.time() - t_iter)
    return {"u": u, "v": v, "alpha": alpha, "criterions": torch.stack(criterions),
            "mm_iters": mm_iters, "seconds": time.time() - t0, "seconds_mm": seconds_mm, "seconds_iter": seconds_iter}


def clustering_accuracy(u, x_q, y_q, n_class, graph_matching=True):
    """Zero-shot accuracy tail (src/methods/zero_shot/em_dirichlet.py:61-92 with
    src/utils.py:380-417): prototype of each predicted cluster = mean raw feature of its
    members, clusters (in first-appearance order) assigned to classes by minimum-cost
    matching on -prototype (scipy Hungarian) or by plain argmax.  Returns (acc (N,1), new_preds)."""
    import numpy as np
    from scipy.optimize import linear_sum_assignment

    preds = u.argmax(2)
    hot = one_hot_rows(preds, n_class)
    protos = ((hot.unsqueeze(-1) * x_q.unsqueeze(2)).sum(1)) / (hot.sum(1).clamp(min=EPS).unsqueeze(-1))
    sizes = hot.sum(1).unsqueeze(-1)
    protos = protos * (sizes > EPS)
    new_preds = torch.zeros_like(preds)
    for n in range(preds.shape[0]):
        if graph_matching:
            order = []
            for c in preds[n].tolist():
                if c not in order:
                    order.append(c)
            cost = np.zeros((len(order), n_class))
            for i, c in enumerate(order):
                cost[i, :] = -protos[n, c].numpy()
            _, cls = linear_sum_assignment(cost, maximize=False)
            lut = {c: int(cls[i]) for i, c in enumerate(order)}
            new_preds[n] = torch.tensor([lut[c] for c in preds[n].tolist()])
        else:
            new_preds[n] = protos[n].argmax(dim=-1)[preds[n]]
    acc = (new_preds == y_q).float().mean(1, keepdim=True)
    return acc, new_preds


def run_soft_kmeans(x_q, *, n_class, iters, temperature):
    """SOFT_KMEANS on probability features, the reference's torch op sequence
    (src/methods/zero_shot/soft_kmeans.py:105-220): w = u^T z / sum u (empty clusters keep their
    centroid), u = softmax_k(-T/2 ||w_k - z_q||^2).  Returns dict(u, w, criterions, seconds);
    the logged criterion is identically 0 (the reference compares u with a copy of itself)."""
    query = x_q.clone().float()
    t0 = time.time()
    u = query.clone()
    num = (query.unsqueeze(2) * u.unsqueeze(3)).sum(1)
    den = u.sum(1).clamp(min=EPS)
    w = num.div_(den.unsqueeze(2))
    criterions = []
    for _ in range(iters):
        num = (query.unsqueeze(2) * u.unsqueeze(3)).sum(1)
        den = u.sum(1).clamp(min=EPS)
        live = u.sum(1).unsqueeze(-1) > EPS
        w = num.div_(den.unsqueeze(2)) * live + (w * (1 - 1 * live))
        diff = w.unsqueeze(1) - query.unsqueeze(2)
        logits = -1 / 2 * (diff.square_()).sum(dim=-1)
        u = (temperature * logits).softmax(2)
        criterions.append((u.clone() - u).norm(dim=(1, 2)).mean(0))
    return {"u": u, "w": w, "criterions": torch.stack(criterions), "seconds": time.time() - t0}


def run_hard_kmeans(x_q, *, n_class, iters):
    """HARD_KMEANS on probability features, the reference's torch op sequence
    (src/methods/zero_shot/hard_kmeans.py:26-35, 128-152, 186-204): w = u^T z / sum u with empty
    clusters set to ZERO, u = one_hot(argmin_k softmax_k(||w_k - z_q||^2)) - the softmax is kept
    because ties after its rounding decide the argmin.  Returns dict(u, w, criterions (iters,),
    labels (iters,N,Q), seconds); the reference logs every criterion twice."""
    query = x_q.clone().float()
    t0 = time.time()
    u = query.clone()
    u_old = u.clone()
    criterions, labels_all = [], []
    w = None
    for _ in range(iters):
        num = (query.unsqueeze(2) * u.unsqueeze(3)).sum(1)
        den = u.sum(1).clamp(min=EPS)
        live = u.sum(1).unsqueeze(-1) > EPS
        w = num.div_(den.unsqueeze(2)) * live
        diff = w.unsqueeze(1) - query.unsqueeze(2)
        logits = (diff.square_()).sum(dim=-1)
        labels = torch.argmin(logits.softmax(2), dim=-1)
        u = torch.zeros_like(u)
        u.scatter_(2, labels.unsqueeze(-1), 1.0)
        labels_all.append(labels.clone())
        criterions.append((u_old - u).norm(dim=(1, 2)).mean(0))
        u_old = u.clone()
    return {"u": u, "w": w, "criterions": torch.stack(criterions), "labels": torch.stack(labels_all),
            "seconds": time.time() - t0}


def run_paddle(x_q, x_s, y_s, *, n_class, iters, lambd):
    """PADDLE on probability features, the reference's torch op sequence
    (src/methods/few_shot/paddle.py:94-219): prototypes from the support class means, then
    u = softmax_k(-1/2 ||w_k - z_q||^2 + lambd v_k / Q), v = log(mean_q u + eps) + 1,
    w = (sum_q u z + support sums) / (sum_q u + support counts).  Returns dict(u, v, w, criterions,
    argmax (iters,N,Q), seconds); the logged criterion is identically 0."""
    query, support = x_q.clone().float(), x_s.clone().float()
    n_task, n_query = query.shape[0], query.shape[1]
    t0 = time.time()
    v = torch.zeros(n_task, n_class)
    ys_hot = one_hot_rows(y_s.long().view(n_task, -1), n_class)
    counts = ys_hot.sum(1).unsqueeze(-1)
    w = (ys_hot.unsqueeze(-1) * support.unsqueeze(2)).sum(1).div_(counts)
    criterions, argmax = [], []
    u = query.clone()
    for _ in range(iters):
        diff = w.unsqueeze(1) - query.unsqueeze(2)
        logits = -1 / 2 * (diff.square_()).sum(dim=-1)
        u = (logits + lambd * v.unsqueeze(1) / n_query).softmax(2)
        argmax.append(u.argmax(2).clone())
        v = torch.log(u.sum(1) / u.size(1) + EPS) + 1
        num = (query.unsqueeze(2) * u.unsqueeze(3)).sum(1)
        den = u.sum(1)
        num.add_((support.unsqueeze(2) * ys_hot.unsqueeze(3)).sum(1))
        den.add_(ys_hot.sum(1))
        w = num.div_(den.unsqueeze(2))
        criterions.append((u.clone() - u).norm(dim=(1, 2)).mean(0))
    return {"u": u, "v": v, "w": w, "criterions": torch.stack(criterions), "argmax": torch.stack(argmax),
            "seconds": time.time() - t0}


def run_em_gaussian(x_q, *, n_class, iters, temperature, lambd):
    """EM_GAUSSIAN on probability features, the reference's torch op sequence
    (src/methods/zero_shot/em_gaussian.py:107-229): SOFT_KMEANS plus the class-proportion term,
    u = softmax_k(T * (-1/2 ||w_k - z_q||^2) + lambd v_k / Q), v = log(mean_q u + eps) + 1.
    Returns dict(u, v, w, criterions, argmax (iters,N,Q), seconds); the logged criterion is 0."""
    query = x_q.clone().float()
    n_task, n_query = query.shape[0], query.shape[1]
    t0 = time.time()
    v = torch.zeros(n_task, n_class)
    u = query.clone()
    num = (query.unsqueeze(2) * u.unsqueeze(3)).sum(1)
    den = u.sum(1).clamp(min=EPS)
    w = num.div_(den.unsqueeze(2))
    criterions, argmax = [], []
    for _ in range(iters):
        num = (query.unsqueeze(2) * u.unsqueeze(3)).sum(1)
        den = u.sum(1).clamp(min=EPS)
        live = u.sum(1).unsqueeze(-1) > EPS
        w = num.div_(den.unsqueeze(2)) * live + (w * (1 - 1 * live))
        diff = w.unsqueeze(1) - query.unsqueeze(2)
        logits = -1 / 2 * (diff.square_()).sum(dim=-1)
        u = (temperature * logits + lambd * v.unsqueeze(1) / n_query).softmax(2)
        argmax.append(u.argmax(2).clone())
        v = torch.log(u.sum(1) / u.size(1) + EPS) + 1
        criterions.append((u.clone() - u).norm(dim=(1, 2)).mean(0))
    return {"u": u, "v": v, "w": w, "criterions": torch.stack(criterions), "argmax": torch.stack(argmax),
            "seconds": time.time() - t0}


def _bdcspn_logits(w, samples):
    """BDCSPN.get_logits (few_shot/bdcspn.py:42-58): -1/2 squared distance of the L2-normalised arguments."""
    w = w / w.norm(p=2, dim=-1, keepdim=True)
    samples = samples / samples.norm(p=2, dim=-1, keepdim=True)
    if len(w.shape) == 3:
        diff = w.unsqueeze(1) - samples.unsqueeze(2)
    else:
        diff = w.unsqueeze(0) - samples.unsqueeze(1)
    return -1 / 2 * (diff.square_()).sum(dim=-1)


def run_bdcspn(x_q, x_s, y_s, *, n_class, temp, norm_type="L2N"):
    """BD-CSPN on probability features, the reference's torch op sequence
    (src/methods/few_shot/bdcspn.py:77-200): feature normalisation (CL2N / L2N / none), support
    class means, per task a query shift eta = mean(support) - mean(query), soft assignment of
    support + shifted queries to the means, rectified prototypes = assignment-weighted means of
    the normalised augmented set, prediction = argmax softmax(temp * -1/2 ||.||^2) against them.
    Returns dict(prototypes (N,K,C), u (N,Q,K), preds (N,Q), seconds)."""
    support, query = x_s.clone().float(), x_q.clone().float()
    y_s = y_s.long().view(support.shape[0], -1)
    t0 = time.time()
    train_mean = support.mean(1).unsqueeze(1)
    if norm_type == "CL2N":
        support = support - train_mean
        support = support / support.norm(p=2, dim=2, keepdim=True)
        query = query - train_mean
        query = query / query.norm(p=2, dim=2, keepdim=True)
    elif norm_type == "L2N":
        support = support / support.norm(p=2, dim=2, keepdim=True)
        query = query / query.norm(p=2, dim=2, keepdim=True)
    n_task, n_query, dim = query.shape
    prototypes = torch.zeros(n_task, n_class, dim)
    ys_hot = one_hot_rows(y_s, n_class)
    counts = ys_hot.sum(1).unsqueeze(-1)
    init = (ys_hot.unsqueeze(-1) * support.unsqueeze(2)).sum(1).div_(counts)
    for j in range(n_task):
        eta = support[j].mean(0) - query[j].mean(0)
        aug = torch.cat((support[j], query[j] + eta), dim=0)
        u = (temp * _bdcspn_logits(init[j], aug)).softmax(-1)
        aug = aug / aug.norm(p=2, dim=-1, keepdim=True)
        cnt = u.sum(0).unsqueeze(-1)
        prototypes[j] = (u.unsqueeze(-1) * aug.unsqueeze(1)).sum(0).div_(cnt)
    u = (temp * _bdcspn_logits(prototypes, query)).softmax(-1)
    return {"prototypes": prototypes, "u": u, "preds": u.argmax(2), "seconds": time.time() - t0}


def run_em_gaussian_cov(x_q, *, n_class, iters, lambd):
    """EM_GAUSSIAN_COV on probability features, the reference's torch op sequence
    (src/methods/zero_shot/em_gaussian_cov.py:106-257): EM_GAUSSIAN with a diagonal inverse
    covariance s per cluster, s = sum_q u / clamp(sum_q u (w - z_q)^2, eps),
    u = softmax_k(-1/2 sum_d s (w - z)^2 + 1/2 sum_d log(s + eps) + lambd v_k / Q); no temperature.
    Returns dict(u, v, w, s, criterions, argmax (iters,N,Q), seconds)."""
    query = x_q.clone().float()
    n_task, n_query = query.shape[0], query.shape[1]
    t0 = time.time()
    v = torch.zeros(n_task, n_class)
    u = query.clone()
    num = (query.unsqueeze(2) * u.unsqueeze(3)).sum(1)
    den = u.sum(1).clamp(min=EPS)
    w = num.div_(den.unsqueeze(2))
    d_q = ((w.unsqueeze(1) - query.unsqueeze(2)).square_()).mul_(u.unsqueeze(3)).sum(1)
    s = (u.sum(1)).unsqueeze(2) / d_q.clamp(min=EPS)
    criterions, argmax = [], []
    for _ in range(iters):
        num = (query.unsqueeze(2) * u.unsqueeze(3)).sum(1)
        den = u.sum(1).clamp(min=EPS)
        live = u.sum(1).unsqueeze(-1) > EPS
        w = num.div_(den.unsqueeze(2)) * live + (w * (1 - 1 * live))
        d_q = ((w.unsqueeze(1) - query.unsqueeze(2)).square_()).mul_(u.unsqueeze(3)).sum(1)
        s = (u.sum(1)).unsqueeze(2) / d_q.clamp(min=EPS) * live + (s * (1 - 1 * live))
        diff = w.unsqueeze(1) - query.unsqueeze(2)
        logits = -1 / 2 * ((diff.square_()).mul_(s.unsqueeze(1))).sum(dim=-1)
        det = 1 / 2 * (torch.log(s + EPS).sum(-1)).unsqueeze(1)
        u = (logits + det + lambd * v.unsqueeze(1) / n_query).softmax(2)
        argmax.append(u.argmax(2).clone())
        v = torch.log(u.sum(1) / u.size(1) + EPS) + 1
        criterions.append((u.clone() - u).norm(dim=(1, 2)).mean(0))
    return {"u": u, "v": v, "w": w, "s": s, "criterions": torch.stack(criterions), "argmax": torch.stack(argmax),
            "seconds": time.time() - t0}


def run_kl_kmeans(x_q, *, n_class, iters):
    """KL_KMEANS on probability features, the reference's torch op sequence
    (src/methods/zero_shot/kl_kmeans.py:123-189): centroids w = (u^T z) / max(sum u, 1) (a bmm),
    zero for empty clusters; every query goes to the centroid of smallest
    KL(z + eps || w + eps) = sum_d P log(P / Q).  Returns dict(u, w, criterions (iters,),
    labels (iters,N,Q), seconds); the reference logs every criterion twice."""
    query = x_q.clone().float()
    t0 = time.time()
    u = query.clone()
    u_old = u.clone()
    criterions, labels_all = [], []
    w = None
    for _ in range(iters):
        cluster_sizes = u.sum(1).unsqueeze(-1)
        nonzero = cluster_sizes > 0
        w = (u.transpose(1, 2) @ query) / cluster_sizes.clamp(min=1)
        w *= nonzero.float()
        P = query.unsqueeze(2) + EPS
        Qm = w.unsqueeze(1) + EPS
        divs = torch.sum(P * torch.log(P / Qm), dim=-1)
        labels = torch.argmin(divs, dim=-1)
        u = torch.zeros_like(u)
        u.scatter_(2, labels.unsqueeze(-1), 1.0)
        labels_all.append(labels.clone())
        criterions.append((u_old - u).norm(dim=(1, 2)).mean(0))
        u_old = u.clone()
    return {"u": u, "w": w, "criterions": torch.stack(criterions), "labels": torch.stack(labels_all),
            "seconds": time.time() - t0}


def run_alpha_tim(x_q, x_s, y_s, *, n_class, iters, temp, lr, alpha_value, loss_weights=(1.0, 1.0, 1.0),
                  entropies=("Shannon", "Alpha", "Alpha")):
    """ALPHA_TIM, the reference's torch op sequence with autograd and torch.optim.Adam
    (src/methods/few_shot/tim.py:192-322): weights = support class means, then `iters` Adam steps on
    lw0 * ce - (lw1 * q_ent - lw2 * q_cond_ent) with logits temp * (x w^T - |w|^2/2 - |x|^2/2).
    Returns dict(weights, logits_q (last iteration's forward pass), criterions (iters,), argmax (N,Q), seconds)."""
    support, query = x_s.clone().float(), x_q.clone().float()
    n_task = query.shape[0]
    y_s = y_s.long().view(n_task, -1)
    t0 = time.time()
    ys_hot = one_hot_rows(y_s, n_class)
    counts = ys_hot.sum(1).view(n_task, -1, 1)
    weights = (ys_hot.transpose(1, 2).matmul(support) / counts).requires_grad_()

    def get_logits(samples):
        return temp * (samples.matmul(weights.transpose(1, 2)) - 1 / 2 * (weights ** 2).sum(2).view(n_task, 1, -1)
                       - 1 / 2 * (samples ** 2).sum(2).view(n_task, -1, 1))

    optimizer = torch.optim.Adam([weights], lr=lr)
    lw, a, criterions, logits_q = list(loss_weights), alpha_value, [], None
    for _ in range(iters):
        weights_old = weights.detach().clone()
        logits_s, logits_q = get_logits(support), get_logits(query)
        q_probs = logits_q.softmax(2)
        if entropies[0] == "Shannon":
            ce = -(ys_hot * torch.log(logits_s.softmax(2) + 1e-12)).sum(2).mean(1).sum(0)
        elif entropies[0] == "Alpha":
            ce = torch.pow(ys_hot, a) * torch.pow(logits_s.softmax(2) + 1e-12, 1 - a)
            ce = ((1 - ce.sum(2)) / (a - 1)).mean(1).sum(0)
        else:
            raise ValueError("Entropies must be in ['Shannon', 'Alpha']")
        if entropies[1] == "Shannon":
            q_ent = -(q_probs.mean(1) * torch.log(q_probs.mean(1))).sum(1).sum(0)
        elif entropies[1] == "Alpha":
            q_ent = ((1 - (torch.pow(q_probs.mean(1), a)).sum(1)) / (a - 1)).sum(0)
        else:
            raise ValueError("Entropies must be in ['Shannon', 'Alpha']")
        if entropies[2] == "Shannon":
            q_cond_ent = -(q_probs * torch.log(q_probs + 1e-12)).sum(2).mean(1).sum(0)
        elif entropies[2] == "Alpha":
            q_cond_ent = ((1 - (torch.pow(q_probs + 1e-12, a)).sum(2)) / (a - 1)).mean(1).sum(0)
        else:
            raise ValueError("Entropies must be in ['Shannon', 'Alpha']")
        loss = lw[0] * ce - (lw[1] * q_ent - lw[2] * q_cond_ent)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        criterions.append((weights_old - weights.detach()).norm(dim=-1).mean())
    logits_q = logits_q.detach()
    return {"weights": weights.detach(), "logits_q": logits_q, "criterions": torch.stack(criterions),
            "argmax": logits_q.argmax(2), "seconds": time.time() - t0}


def run_laplacian_shot(x_q, x_s, y_s, y_q, *, n_class, iters, knn, lmd, norm_type="L2N"):
    """LAPLACIAN_SHOT, the reference's numpy / scipy.sparse / sklearn op sequence
    (src/methods/few_shot/laplacian_shot.py:66-249) with `dtype=float` where the reference writes the removed alias
    `np.float` (:100).  Returns dict(unary (N,Q,K) f32, neighbours (N,Q,knn-1) sorted, preds (N,Q), acc (N,iters),
    ent_energy (N,iters) f64)."""
    import numpy as np
    from numpy import linalg as LA
    from scipy import sparse
    from sklearn.neighbors import NearestNeighbors
    z_s, z_q = x_s.clone().float().cpu(), x_q.clone().float().cpu()
    if norm_type == "L2N":
        # the reference divides the torch tensor by LA.norm's numpy result (:84-85); same numbers without numpy's
        # __array_wrap__ deprecation warning
        z_s = z_s / torch.from_numpy(LA.norm(z_s.numpy(), 2, 2))[:, :, None]
        z_q = z_q / torch.from_numpy(LA.norm(z_q.numpy(), 2, 2))[:, :, None]
    n_task = z_q.shape[0]
    y_s = y_s.long().view(n_task, -1)
    y_q = y_q.long().view(n_task, -1).numpy()
    one_hot = one_hot_rows(y_s, n_class)
    counts = one_hot.sum(1).view(n_task, -1, 1)
    support = (one_hot.transpose(1, 2).matmul(z_s) / counts).numpy()
    query = z_q.numpy()

    def normalize(y_in):
        y_in = y_in - np.max(y_in, axis=1)[:, np.newaxis]
        y_out = np.exp(y_in)
        return y_out / (np.sum(y_out, axis=1)[:, None])

    out = {"unary": [], "neighbours": [], "preds": [], "acc": [], "ent_energy": []}
    for i in range(n_task):
        distance = LA.norm(support[i][:, None, :] - query[i], 2, axis=-1)
        unary = distance.transpose() ** 2
        n = query[i].shape[0]
        _, knnind = NearestNeighbors(n_neighbors=knn).fit(query[i]).kneighbors(query[i])
        row, col = np.repeat(range(n), knn - 1), knnind[:, 1:].flatten()
        kernel = sparse.csc_matrix((np.ones(n * (knn - 1)), (row, col)), shape=(n, n), dtype=float)
        old_e, y, energies, accs, pred = float("inf"), normalize(-unary), [], [], None
        for it in range(iters):
            y = normalize(-unary - (-lmd * kernel.dot(y)))
            pairwise = kernel.dot(y)
            e = (y * np.log(np.maximum(y, 1e-20)) + ((unary * y) + (-lmd * pairwise * y))).sum()
            energies.append(e)
            pred = np.argmax(y, axis=1)
            accs.append(np.float32((y_q[i] == pred).astype(np.float32).mean()))
            if it > 1 and abs(e - old_e) <= 1e-6 * abs(old_e):
                energies += [e] * (iters - it - 1)
                accs += [accs[-1]] * (iters - it - 1)
                break
            old_e = e
        out["unary"].append(unary.astype(np.float32))
        out["neighbours"].append(np.sort(knnind[:, 1:], axis=1))
        out["preds"].append(pred)
        out["acc"].append(accs)
        out["ent_energy"].append(energies)
    return {k: np.asarray(v) for k, v in out.items()}
