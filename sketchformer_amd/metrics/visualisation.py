"""tsne / tsne-predicted / pca projections of the embeddings (metrics/visualisation.py of the reference): up to 1000
samples of the first 10 classes met in a seeded permutation; rows of the result = (x, y, label)."""
import numpy as np

from ..core.metrics import ProjectionMetric


def _select(y, pred_z, shown_labels):
    np.random.seed(14)
    idx = np.random.permutation(len(y))
    np.random.seed()
    y, pred_z, shown_labels = np.asarray(y).reshape(-1)[idx], np.asarray(pred_z)[idx], np.asarray(shown_labels).reshape(-1)[idx]
    feats, labels, chosen = [], [], []
    for label, feature, shown in zip(y, pred_z, shown_labels):
        if label not in chosen and len(chosen) < 10:
            chosen.append(label)
        if label in chosen:
            feats.append(np.asarray(feature).reshape(-1))
            labels.append(shown)
        if len(feats) >= 1000:
            break
    return np.array(feats), np.array(labels, dtype=np.float64)


def _tsne(feats):
    from sklearn.manifold import TSNE
    return TSNE(n_components=2, verbose=0, perplexity=min(30, max(2, len(feats) // 4)), random_state=14).fit_transform(feats)


class TSNEProjection(ProjectionMetric):
    name = 'tsne'
    input_type = 'predictions_on_validation_set'

    def compute(self, input_data):
        x, y, pred_x, pred_y, pred_z, tokenizer, plot_filepath, tmp_filepath, _ = input_data
        feats, labels = _select(y, pred_z, y)
        return np.concatenate((_tsne(feats), labels[:, None]), axis=1)


class PredictedLabelsTSNEProjection(ProjectionMetric):
    name = 'tsne-predicted'
    input_type = 'predictions_on_validation_set'

    def compute(self, input_data):
        x, y, pred_x, pred_y, pred_z, tokenizer, plot_filepath, tmp_filepath, _ = input_data
        feats, labels = _select(y, pred_z, pred_y)
        return np.concatenate((_tsne(feats), labels[:, None]), axis=1)


class PCAProjection(ProjectionMetric):
    name = 'pca'
    input_type = 'predictions_on_validation_set'

    def compute(self, input_data):
        from sklearn.decomposition import PCA
        x, y, pred_x, pred_y, pred_z, tokenizer, plot_filepath, tmp_filepath, _ = input_data
        feats, labels = _select(y, pred_z, y)
        return np.concatenate((PCA(n_components=2).fit_transform(feats), labels[:, None]), axis=1)
