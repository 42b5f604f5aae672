"""Flatten-then-reduce wrapper for sklearn pipelines — counterpart of the reference's
decomposition/DimRedReshape.py:11-78 (config 1: ``make_pipeline(DimRedReshape(PCA), SVC(...))``)."""
from sklearn.base import BaseEstimator


class DimRedReshape(BaseEstimator):
    def __init__(self, dim_red, n_components=None):
        self.dim_red = dim_red
        self.n_components = n_components

    def fit(self, X, y=None):
        self.transformer = self.dim_red(n_components=self.n_components)
        self.transformer.fit(X.reshape(X.shape[0], -1))
        return self

    def transform(self, X, y=None):
        return self.transformer.transform(X.reshape(X.shape[0], -1))

    def fit_transform(self, X, y=None):
        return self.fit(X).transform(X)
