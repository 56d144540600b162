// dorado_amd/host/tensor_loader.h — libtorch-free reader of ONT ".tensor" files.
//
// The reference loads model weights (and test signals) with torch::load(std::vector<at::Tensor>&, path)
// (torch_utils/tensor_utils.cpp:153-163), i.e. a TorchScript archive: a ZIP with STORED (uncompressed)
// entries <stem>/data.pkl (pickle protocol 2: a module object whose attributes "0", "1", … are
// torch._utils._rebuild_tensor_v2(storage, offset, size, stride, …) calls) and <stem>/data/<key> (the raw
// little-endian storage).  This reader parses exactly that: ZIP central directory, the pickle opcode subset
// TorchScript emits, and materialises each tensor (honouring storage offset, size and strides).
// File names per layer: load_lstm_model_weights / load_tx_model_weights (basecall/crf_utils.cpp:26-150).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace dorado_amd::host {

enum class DType { F16, BF16, F32, F64, I8, U8, I16, I32, I64, BOOL };

struct LoadedTensor {
    std::string name;               // attribute name inside the archive ("0", "1", …)
    DType dtype = DType::F32;
    std::vector<int64_t> shape;
    std::vector<uint8_t> data;      // contiguous, row-major, native dtype
    size_t numel() const;
    std::vector<float> to_float() const;   // exact for f16 / bf16 / f32 / small ints
};

// All tensors of one archive, in attribute order.  Throws std::runtime_error with the reason on any
// structural problem (not a zip, compressed entry, unsupported pickle opcode, storage out of range …).
std::vector<LoadedTensor> load_tensor_file(const std::string &path);

// File names in module.parameters() order for a model directory — basecall/crf_utils.cpp:26-150.
std::vector<std::string> lstm_model_tensor_names(int n_convs, int lstm_layers, bool flstm, bool linear_bias,
                                                 bool decomposition);
std::vector<std::string> tx_model_tensor_names(int n_convs, int depth);

}  // namespace dorado_amd::host
