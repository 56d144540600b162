// oracle/ref_common.h — TEST INFRASTRUCTURE ONLY.  The model descriptor the Python side hands to the compiled-reference drivers
// (ref_driver.cpp, ref_pipeline.cpp) and the reference configuration object built from it.  Our code; calls into the reference's
// config structs only (dorado/config/include/config/BasecallModelConfig.h).
#pragma once
#include "config/BasecallModelConfig.h"

#include <string>

extern "C" {

struct RefModelDesc {
    // convolutions
    int n_convs;
    int conv_insize[8], conv_size[8], conv_winlen[8], conv_stride[8];
    int conv_act[8];  // 0 swish, 1 swish_clamp, 2 tanh  (config/common.h Activation)
    // LSTM-CRF
    int lstm_size, lstm_layers;
    int state_len, outsize;
    int bias;          // linear bias (pre-v4)
    int clamp;         // clamp +-5
    float scale;       // 5.0 => tanh*5 head
    int out_features;  // -1 = no decomposition
    int num_features;
    // Tx (d_model <= 0 => LSTM model)
    int tx_d_model, tx_nhead, tx_depth, tx_dim_ff, tx_win_upper, tx_win_lower, tx_max_seq_len;
    float tx_deepnorm_alpha, tx_theta;
    int up_size, up_scale_factor;
    float crf_scale, crf_blank_score;
    int crf_expand_blanks;
};

}  // extern "C" (helpers below have C++ linkage)

inline dorado::config::BasecallModelConfig ref_make_config(const RefModelDesc &d) {
    dorado::config::BasecallModelConfig c;
    for (int i = 0; i < d.n_convs; ++i) {
        dorado::config::ConvParams p;
        p.insize = d.conv_insize[i];
        p.size = d.conv_size[i];
        p.winlen = d.conv_winlen[i];
        p.stride = d.conv_stride[i];
        p.activation = static_cast<dorado::config::Activation>(d.conv_act[i]);
        c.convs.push_back(p);
    }
    c.lstm_size = d.lstm_size;
    c.lstm_layers = d.lstm_layers;
    c.state_len = d.state_len;
    c.outsize = d.outsize;
    c.bias = d.bias != 0;
    c.clamp = d.clamp != 0;
    c.scale = d.scale;
    c.blank_score = 2.0f;
    c.num_features = d.num_features;
    if (d.out_features > 0) {
        c.out_features = d.out_features;
    }
    c.stride = 1;
    for (int i = 0; i < d.n_convs; ++i) {
        c.stride *= d.conv_stride[i];
    }
    if (d.tx_d_model > 0) {
        dorado::config::TxStack tx;
        tx.tx.d_model = d.tx_d_model;
        tx.tx.nhead = d.tx_nhead;
        tx.tx.depth = d.tx_depth;
        tx.tx.dim_feedforward = d.tx_dim_ff;
        tx.tx.attn_window = {d.tx_win_upper, d.tx_win_lower};
        tx.tx.deepnorm_alpha = d.tx_deepnorm_alpha;
        tx.tx.theta = d.tx_theta;
        tx.tx.max_seq_len = d.tx_max_seq_len;
        tx.upsample.size = d.up_size;
        tx.upsample.scale_factor = d.up_scale_factor;
        tx.crf.insize = d.up_size;
        tx.crf.n_base = 4;
        tx.crf.state_len = d.state_len;
        tx.crf.scale = d.crf_scale;
        tx.crf.blank_score = d.crf_blank_score;
        tx.crf.expand_blanks = d.crf_expand_blanks != 0;
        tx.crf.permute = {};
        c.tx = tx;
    }
    return c;
}

