import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    x, y = a[k], b[k]
    same = x.shape == y.shape and (x == y).all()
    msg = ''
    if not same and x.shape == y.shape and x.ndim >= 1:
        bad = np.unique(np.nonzero(x != y)[0])
        msg = ' slots differing: %d of %d, first %s' % (len(bad), x.shape[0], bad[:8])
    print(k, 'same' if same else 'DIFF', msg)
