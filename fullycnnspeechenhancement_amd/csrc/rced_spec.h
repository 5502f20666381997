// Layer tables of the three networks, transcribed from the reference's graph builders:
//   V1  FullyCNNSEModel    model_utils/model.py:6-29
//   V2  FullyCNNSEModelV2  model_utils/model.py:32-61
//   V3  FullyCNNSEModelV3  model_utils/model.py:64-96
// op semantics: model_utils/module.py:11-34 (conv SAME stride 1 + bias -> BN -> +skip -> ReLU).
#pragma once
#include <cstddef>

namespace rced {

constexpr int kFeatureDim = 129;   // [data] feature_dim
constexpr float kBnEps = 1e-3f;    // tf.layers.batch_normalization default epsilon
constexpr int kMaxLayers = 16;

struct LayerSpec {
  const char* scope;  // TF variable scope
  int cout, kh, kw;
  int use_norm, use_act;
  int src;        // tensor id read by the conv: 0 = input, k+1 = output of layer k
  int skip_pre;   // tensor id added after BN, before ReLU (module.py:30-31), -1 none
  int skip_post;  // tensor id added after ReLU (V3 block skip, model.py:75-76), -1 none
};

struct NetSpec {
  int n_layers;
  LayerSpec layer[kMaxLayers];
};

// R-CED V1.  The fifth encoder's scope really is "encode_8" (model.py:15).
constexpr NetSpec kV1 = {10, {
  {"encode_1", 12, 8, 13, 1, 1, 0, -1, -1},
  {"encode_2", 16, 1, 11, 1, 1, 1, -1, -1},
  {"encode_3", 20, 1, 9, 1, 1, 2, -1, -1},
  {"encode_4", 24, 1, 7, 1, 1, 3, -1, -1},
  {"encode_8", 32, 1, 7, 1, 1, 4, -1, -1},
  {"decode_1", 24, 1, 7, 1, 1, 5, 4, -1},
  {"decode_2", 20, 1, 9, 1, 1, 6, 3, -1},
  {"decode_3", 16, 1, 11, 1, 1, 7, 2, -1},
  {"decode_4", 12, 1, 13, 1, 1, 8, 1, -1},
  {"decode_5", 1, 1, 129, 0, 0, 9, -1, -1},
}};

// R-CED V2.
constexpr NetSpec kV2 = {16, {
  {"encode_1", 10, 8, 11, 1, 1, 0, -1, -1},
  {"encode_2", 12, 1, 7, 1, 1, 1, -1, -1},
  {"encode_3", 14, 1, 5, 1, 1, 2, -1, -1},
  {"encode_4", 15, 1, 5, 1, 1, 3, -1, -1},
  {"encode_5", 19, 1, 5, 1, 1, 4, -1, -1},
  {"encode_6", 21, 1, 5, 1, 1, 5, -1, -1},
  {"encode_7", 23, 1, 7, 1, 1, 6, -1, -1},
  {"encode_8", 25, 1, 11, 1, 1, 7, -1, -1},
  {"decode_1", 23, 1, 7, 1, 1, 8, 7, -1},
  {"decode_2", 21, 1, 5, 1, 1, 9, 6, -1},
  {"decode_3", 19, 1, 5, 1, 1, 10, 5, -1},
  {"decode_4", 15, 1, 5, 1, 1, 11, 4, -1},
  {"decode_5", 14, 1, 5, 1, 1, 12, 3, -1},
  {"decode_6", 12, 1, 7, 1, 1, 13, 2, -1},
  {"decode_7", 10, 1, 11, 1, 1, 14, 1, -1},
  {"decode_8", 1, 1, 129, 0, 0, 15, -1, -1},
}};

// CR-CED V3: five simple_RCED blocks (18 -> 30 -> 8) + decode_final.
// Block outputs: CE1 = tensor 3, CE2 = tensor 6, CE3 = 9, CD1 = 12 (+CE2), CD2 = 15 (+CE1).
constexpr NetSpec kV3 = {16, {
  {"CE1_encode_1", 18, 8, 9, 1, 1, 0, -1, -1},
  {"CE1_encode_2", 30, 1, 5, 1, 1, 1, -1, -1},
  {"CE1_decode", 8, 1, 9, 1, 1, 2, -1, -1},
  {"CE2_encode_1", 18, 1, 9, 1, 1, 3, -1, -1},
  {"CE2_encode_2", 30, 1, 5, 1, 1, 4, -1, -1},
  {"CE2_decode", 8, 1, 9, 1, 1, 5, -1, -1},
  {"CE3_encode_1", 18, 1, 9, 1, 1, 6, -1, -1},
  {"CE3_encode_2", 30, 1, 5, 1, 1, 7, -1, -1},
  {"CE3_decode", 8, 1, 9, 1, 1, 8, -1, -1},
  {"CD1_encode_1", 18, 1, 9, 1, 1, 9, -1, -1},
  {"CD1_encode_2", 30, 1, 5, 1, 1, 10, -1, -1},
  {"CD1_decode", 8, 1, 9, 1, 1, 11, -1, 6},
  {"CD2_encode_1", 18, 1, 9, 1, 1, 12, -1, -1},
  {"CD2_encode_2", 30, 1, 5, 1, 1, 13, -1, -1},
  {"CD2_decode", 8, 1, 9, 1, 1, 14, -1, 3},
  {"decode_final", 1, 1, 129, 0, 0, 15, -1, -1},
}};

inline const NetSpec* net_spec(int variant) {
  switch (variant) {
    case 1: return &kV1;
    case 2: return &kV2;
    case 3: return &kV3;
    default: return nullptr;
  }
}

inline int layer_cin(const NetSpec& n, int i) {
  const int s = n.layer[i].src;
  return s == 0 ? 1 : n.layer[s - 1].cout;
}

inline size_t layer_num_weights(const NetSpec& n, int i) {
  const LayerSpec& l = n.layer[i];
  size_t k = (size_t)l.kh * l.kw * layer_cin(n, i) * l.cout + l.cout;
  if (l.use_norm) k += 4 * (size_t)l.cout;
  return k;
}

inline size_t net_num_weights(const NetSpec& n) {
  size_t k = 0;
  for (int i = 0; i < n.n_layers; ++i) k += layer_num_weights(n, i);
  return k;
}

inline size_t net_num_trainable(const NetSpec& n) {  // as trainer.py:78-84 counts
  size_t k = 0;
  for (int i = 0; i < n.n_layers; ++i) {
    const LayerSpec& l = n.layer[i];
    k += (size_t)l.kh * l.kw * layer_cin(n, i) * l.cout + l.cout + (l.use_norm ? 2 * l.cout : 0);
  }
  return k;
}

}  // namespace rced
