"""feature_column.py -- light shims of the tf.feature_column constructors the reference uses.

They carry only what the hot path needs (name, key, num_buckets, dimension, combiner, boundaries,
vocabulary) and turn a raw feature into int64 ids:
  identity  : id in [0, num_buckets) else default_value          models/LFM/Mixture_1/train.py:34-37
  hash      : FarmHash Fingerprint64(str(x)) mod buckets         models/DeepCrossNetwork/train.py:85-86
  vocabulary: index in the list, OOV -> -1 (dropped by the bag)  models/DeepCrossNetwork/train.py:63-82
  bucketized: number of boundaries <= x                          models/DeepFM/deepFM.py:95 (docstring)
Column names follow [TF-upstream] naming (`<key>_embedding`, `<key>_indicator`, `<key>_bucketized`) because
DCN's input_layer concatenates columns sorted by name (DeepCrossNetwork.py:126).
"""
from collections import namedtuple

import torch

from . import ops

# A variable-length (multi-hot) feature: CSR over the batch, the dense stand-in for the SparseTensor a
# TF categorical column receives.  values [nnz] (ids or raw keys), offsets [B+1] int64, weights [nnz]|None.
Ragged = namedtuple("Ragged", ["values", "offsets", "weights"], defaults=[None])


class _Column:
    is_dense = False
    is_categorical = False


class NumericColumn(_Column):
    is_dense = True

    def __init__(self, key, shape=(1,)):
        self.key = key
        self.name = key
        self.shape = tuple(shape)
        self.dimension = 1
        for s in self.shape:
            self.dimension *= int(s)


class _Categorical(_Column):
    is_categorical = True
    weight_key = None

    def ids(self, features, device):
        """-> one-hot LongTensor [B]  or  multi-hot (values [nnz], offsets [B+1], weights|None)."""
        raw = features[self.key]
        if isinstance(raw, Ragged):
            vals = self._to_ids(raw.values, device)
            w = raw.weights
            return vals, raw.offsets.to(device=device, dtype=torch.int64), (w.to(device) if w is not None else None)
        t = self._to_ids(raw, device)
        return t.reshape(-1)

    def _to_ids(self, raw, device):
        raise NotImplementedError


class IdentityCategoricalColumn(_Categorical):
    def __init__(self, key, num_buckets, default_value=None):
        if num_buckets is None or num_buckets < 1:
            raise ValueError("num_buckets {} < 1, column_name {}".format(num_buckets, key))
        if default_value is not None and not (0 <= default_value < num_buckets):
            raise ValueError("default_value {} not in range [0, {}), column_name {}".format(default_value, num_buckets, key))
        self.key = key
        self.name = key
        self.num_buckets = int(num_buckets)
        self.default_value = default_value

    @property
    def range_checked(self):
        """Without a default_value TensorFlow asserts 0 <= id < num_buckets (see _input.check_id_range)."""
        return self.default_value is None

    def _to_ids(self, raw, device):
        t = raw.to(device=device, dtype=torch.int64)
        if self.default_value is not None:
            bad = (t < 0) | (t >= self.num_buckets)
            t = torch.where(bad, torch.full_like(t, int(self.default_value)), t)
        return t


class HashedCategoricalColumn(_Categorical):
    def __init__(self, key, hash_bucket_size):
        if hash_bucket_size is None or hash_bucket_size < 1:
            raise ValueError("hash_bucket_size must be at least 1")
        self.key = key
        self.name = key
        self.num_buckets = int(hash_bucket_size)

    def _to_ids(self, raw, device):
        if isinstance(raw, torch.Tensor):  # integer keys: hashed on the device as decimal text
            return ops.hash_bucket_ints(raw.to(device=device, dtype=torch.int64).reshape(-1), self.num_buckets)
        return ops.hash_bucket_strings(list(raw), self.num_buckets).to(device)


class VocabularyListCategoricalColumn(_Categorical):
    def __init__(self, key, vocabulary_list, default_value=-1):
        self.key = key
        self.name = key
        self.vocabulary = {v: i for i, v in enumerate(vocabulary_list)}
        self.num_buckets = len(self.vocabulary)
        self.default_value = default_value

    def _to_ids(self, raw, device):
        if isinstance(raw, torch.Tensor):
            raw = raw.reshape(-1).tolist()
        return torch.tensor([self.vocabulary.get(v, self.default_value) for v in raw], dtype=torch.int64, device=device)


class BucketizedColumn(_Categorical):
    def __init__(self, source_column, boundaries):
        self.key = source_column.key
        self.name = source_column.key + "_bucketized"
        self.boundaries = [float(b) for b in boundaries]
        if sorted(self.boundaries) != self.boundaries:
            raise ValueError("boundaries must be sorted")
        self.num_buckets = len(self.boundaries) + 1
        self._bd = None

    def _to_ids(self, raw, device):
        if self._bd is None or self._bd.device != torch.device(device):
            self._bd = torch.tensor(self.boundaries, dtype=torch.float32, device=device)
        return ops.bucketize(raw.to(device=device, dtype=torch.float32).reshape(-1), self._bd)


class WeightedCategoricalColumn(_Categorical):
    range_checked = property(lambda self: getattr(self.categorical_column, "range_checked", False))

    """weighted_categorical_column (dataset/SequenceTensorFlowDataset/test4.py:53): ids from the wrapped
    column, per-entry weights from features[weight_feature_key] (same ragged layout)."""

    def __init__(self, categorical_column, weight_feature_key):
        self.categorical_column = categorical_column
        self.key = categorical_column.key
        self.weight_key = weight_feature_key
        self.name = "%s_weighted_by_%s" % (categorical_column.name, weight_feature_key)
        self.num_buckets = categorical_column.num_buckets

    def ids(self, features, device):
        got = self.categorical_column.ids(features, device)
        if not isinstance(got, tuple):
            raise ValueError("weighted_categorical_column needs a Ragged (multi-hot) feature")
        vals, offs, _ = got
        w = features[self.weight_key]
        if isinstance(w, Ragged):
            w = w.values
        return vals, offs, w.to(device=device, dtype=torch.float32).reshape(-1)


class EmbeddingColumn(_Column):
    is_dense = True

    def __init__(self, categorical_column, dimension, combiner="mean", max_norm=None):
        if dimension is None or dimension < 1:
            raise ValueError("Invalid dimension {}.".format(dimension))
        if combiner not in ("mean", "sqrtn", "sum"):
            raise ValueError("combiner must be one of mean, sqrtn, sum")
        self.categorical_column = categorical_column
        self.dimension = int(dimension)
        self.combiner = combiner
        self.max_norm = max_norm          # [TF-upstream] embedding_column(max_norm=): l2-clip every looked-up row
        self.name = categorical_column.name + "_embedding"
        self.num_buckets = categorical_column.num_buckets


class IndicatorColumn(_Column):
    is_dense = True

    def __init__(self, categorical_column):
        self.categorical_column = categorical_column
        self.dimension = categorical_column.num_buckets
        self.name = categorical_column.name + "_indicator"


def numeric_column(key, shape=(1,)):
    return NumericColumn(key, shape)


def categorical_column_with_identity(key, num_buckets, default_value=None):
    return IdentityCategoricalColumn(key, num_buckets, default_value)


def categorical_column_with_hash_bucket(key, hash_bucket_size, dtype=None):
    return HashedCategoricalColumn(key, hash_bucket_size)


def categorical_column_with_vocabulary_list(key, vocabulary_list, dtype=None, default_value=-1):
    return VocabularyListCategoricalColumn(key, vocabulary_list, default_value)


def bucketized_column(source_column, boundaries):
    return BucketizedColumn(source_column, boundaries)


def weighted_categorical_column(categorical_column, weight_feature_key):
    return WeightedCategoricalColumn(categorical_column, weight_feature_key)


def embedding_column(categorical_column, dimension, combiner="mean", max_norm=None):
    return EmbeddingColumn(categorical_column, dimension, combiner, max_norm)


def indicator_column(categorical_column):
    return IndicatorColumn(categorical_column)
