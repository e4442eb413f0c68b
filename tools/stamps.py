"""In-kernel timeline of workgroup 0 of the record-driven kernels (development).

    python tools/stamps.py [fwd|data|filter] [--wave W ...]      (fwd: ring-major forward; data / filter: the backward kernels)

Runs the config-2 layer a few times, then once with the stamp buffer armed (fc_debug_stamp_buffer), and prints for the
chosen wavefronts the cycles between consecutive stamps, labelled as in the kernel source."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

LABELS_FWD = {0: 'run gathered (2 targets)', 1: 'rows converted', 2: 'barrier: slab full', 3: 'contracted', 4: 'barrier: slab free',
              5: 'last slab contracted, partials stored', 6: 'y stored', 10: 'tile start'}


LABELS_DATA = {10: 'tile start', 0: 'gathered', 1: 'slab converted + kept', 2: 'barrier: slab', 3: 'contracted', 4: 'barrier: partials', 5: 'gx terms'}
LABELS_STREAM = {10: 'record start', 0: 'my DMA landed', 2: 'barrier', 4: 'next DMA + gxt reduce', 1: 'next x~ operand', 3: 'products'}
LABELS_FILTER = {10: 'tile start', 1: 'row regrouped', 0: 'x~ operand + next rows requested', 2: 'barrier', 3: 'MFMAs'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('which', nargs='?', default='fwd')
    ap.add_argument('--wave', type=int, nargs='*', default=[0, 7])
    ap.add_argument('--tiles', type=int, default=2)
    ap.add_argument('--warm', type=int, default=5, help='steps before the stamped one (a few hundred: the settled clock)')
    ap.add_argument('--shape', type=int, nargs=5, default=[20000, 32, 48, 2, 6], metavar=('N', 'k', 'C', 'B', 'R'), help='mesh and layer (default: config 2; config 5: 4999 28 64 3 6)')
    args = ap.parse_args()
    if args.which in ('data', 'filter', 'stream'):                # which of the two backward kernels stamps (read once by the library)
        os.environ['FC_STAMP_KERNEL'] = args.which
    if not os.environ.get('FIELDCONV_HIP_LIB'):       # (a development variant built by tools/build_variants.sh)
        from fieldconv_amd import build as _b
        if _b.needs_build() or _b.dev_needs_build():
            __graft_entry__.build()
        os.environ['FIELDCONV_HIP_LIB'] = _b.DEV_LIB_PATH     # the stamps exist in the development build only
    from fieldconv_amd import _lib
    from fieldconv_amd.data import sphere_support
    from fieldconv_amd.nn import FieldConv
    from fieldconv_amd.transforms import FCPrecomp
    dev = torch.device('cuda:0')
    N, k, C, B, R = args.shape
    data = sphere_support(N, k, support='p95').to(dev)
    edges, sten, _, _ = FCPrecomp(B, R, data.epsilon)(data)
    conv = FieldConv(C, C, band_limit=B, n_rings=R).to(dev)
    x = torch.randn(N, C, dtype=torch.cfloat, device=dev).requires_grad_(True)
    gy = torch.randn(N, C, dtype=torch.cfloat, device=dev)
    params = list(conv.parameters())

    def step():
        y = conv(x, edges, sten)
        torch.autograd.grad(y, [x] + params, grad_outputs=gy)
    for _ in range(args.warm):
        step()
    torch.cuda.synchronize()
    buf = torch.zeros(16 * 256, dtype=torch.int64, device=dev)
    lib = _lib.load()
    if args.which == 'fwd':
        lib.fc_debug_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
        with torch.no_grad():
            conv(x, edges, sten)
    else:
        y = conv(x, edges, sten)
        torch.cuda.synchronize()
        lib.fc_debug_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))      # (armed after the forward pass: only the backward kernels stamp)
        torch.autograd.grad(y, [x] + params, grad_outputs=gy)
    torch.cuda.synchronize()
    lib.fc_debug_stamp_buffer(None)
    st = buf.cpu().view(16, 256)
    labels = LABELS_FWD if args.which == 'fwd' else LABELS_FILTER if args.which == 'filter' else LABELS_STREAM if args.which == 'stream' else LABELS_DATA
    t0 = min([int(st[w, 0]) & ((1 << 56) - 1) for w in range(16) if int(st[w, 0])] or [0])
    # per-wave totals by phase (cycles spent BEFORE each label), over the whole launch
    import collections
    print('per-wave cycles by phase (whole launch):')
    names = sorted(labels)
    print('wave ' + ' '.join(f'{labels[k][:14]:>15s}' for k in names) + '   total')
    for w in range(16):
        acc = collections.Counter()
        prev = None
        for v in st[w].tolist():
            if v == 0:
                break
            lab, t = (v >> 56) & 0xff, v & ((1 << 56) - 1)
            if lab in (29, 31):
                continue
            if prev is not None:
                acc[lab] += t - prev
            prev = t
        print(f'{w:4d} ' + ' '.join(f'{acc[k]:15d}' for k in names) + f'  {sum(acc.values()):7d}')
    # clock of the launch: shader cycles (28 -> 30) over the constant 100 MHz counter (29 -> 31)
    ev = {(int(v) >> 56) & 0xff: int(v) & ((1 << 56) - 1) for v in st[0].tolist() if v}
    if all(k in ev for k in (28, 29, 30, 31)) and ev[31] > ev[29]:
        print(f'workgroup 0, wave 0: {ev[30] - ev[28]} shader cycles in {(ev[31] - ev[29]) / 100:.1f} us -> '
              f'{(ev[30] - ev[28]) / (ev[31] - ev[29]) * 0.1:.2f} GHz')
    # from the kernel's first instruction (28) to the first phase label, and from the last phase label to the end (30), per wave
    for w in range(16):
        seq = [((int(v) >> 56) & 0xff, int(v) & ((1 << 56) - 1)) for v in st[w].tolist() if v]
        seq = [(lab, t) for lab, t in seq if lab not in (29, 31)]
        if len(seq) > 3 and seq[0][0] == 28:
            tail = seq[-1][1] - seq[-2][1] if seq[-1][0] == 30 else 0
            first = next(i for i, (lab, _) in enumerate(seq) if lab == 10)
            steps = ' '.join(f'{lab}:+{seq[i][1] - seq[i - 1][1]}' for i, (lab, _) in enumerate(seq[:first + 1]) if i > 0)
            print(f'wave {w:2d}: prologue {seq[first][1] - seq[0][1]:7d} cycles ({steps}), epilogue {tail:7d}')
    for w in args.wave:
        print(f'--- wave {w}')
        prev = None
        tiles = 0
        for v in st[w].tolist():
            if v == 0:
                break
            lab, t = (v >> 56) & 0xff, v & ((1 << 56) - 1)
            if lab in (28, 29, 30, 31):
                continue
            if lab == 10:
                tiles += 1
                if tiles > args.tiles:
                    break
            print(f'  {t - t0:9d}  +{0 if prev is None else t - prev:7d}  {labels.get(lab, lab)}')
            prev = t


if __name__ == '__main__':
    main()
