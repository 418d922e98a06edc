// Plugin interfaces of the scene field and the shader (mirror src/Field/Field.h:10-30, src/Shader/Shader.h:11-22).
#pragma once
#include "GlobalDataPool.h"
#include "Pipe.h"

namespace f2n {

class Field : public Pipe {
 public:
  virtual Tensor Query(const Tensor& coords) { TORCH_CHECK(false, "Not implemented"); return Tensor(); }
  virtual Tensor AnchoredQuery(const Tensor& coords, const Tensor& anchors) { TORCH_CHECK(false, "Not implemented"); return Tensor(); }
  GlobalDataPool* global_data_pool_ = nullptr;
};

class Shader : public Pipe {
 public:
  virtual Tensor Query(const Tensor& feats, const Tensor& dirs) { TORCH_CHECK(false, "Not implemented"); return Tensor(); }
  GlobalDataPool* global_data_pool_ = nullptr;
  int d_in_ = 0, d_out_ = 0;
};

std::unique_ptr<Field> ConstructField(GlobalDataPool* global_data_pool);
std::unique_ptr<Shader> ConstructShader(GlobalDataPool* global_data_pool);

}  // namespace f2n
