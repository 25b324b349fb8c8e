"""Test infrastructure: the scikit-learn pipeline custom_verifier_model.train_verifier_model builds (custom_verifier_model.py:95-113:
FunctionTransformer(flatten) -> StandardScaler -> LogisticRegression), fitted on deterministic random data, and pickled the way the
reference stores it (custom_verifier_model.py:115-117).  A module of its own so that the pickle's flatten function is importable by
the reference (tests/golden/make_golden_onnx.py) and by the tests alike."""
import pickle

import numpy as np


def flatten_features(x):                        # custom_verifier_model.py:91-92
    return [i.flatten() for i in x]


def trained_verifier(seed=0):
    from sklearn.linear_model import LogisticRegression
    from sklearn.pipeline import make_pipeline
    from sklearn.preprocessing import FunctionTransformer, StandardScaler
    r = np.random.default_rng(seed)
    X = r.normal(0, 4.0, (60, 16, 96)).astype(np.float32)
    y = (X[:, :, :8].mean(axis=(1, 2)) > 0).astype(int)
    pipe = make_pipeline(FunctionTransformer(flatten_features), StandardScaler(), LogisticRegression(random_state=0, max_iter=2000, C=0.001))
    pipe.fit(X, y)
    return pipe


def write(path, seed=0):
    with open(path, "wb") as f:
        pickle.dump(trained_verifier(seed), f)
    return path
