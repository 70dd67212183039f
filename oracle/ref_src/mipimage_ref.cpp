// mipimage_ref.cpp — compiles the reference's OWN Source/Renderer/Resources/DXGIUtils.cpp (VQ_DXGI_UTILS::MipImage: the 4-byte box
// filter of material textures and the 16-byte MIN filter of the HDR equirect chain, DXGIUtils.cpp:250-318) from where it lies, against
// the generated stand-ins of mkstubs.py for <dxgiformat.h> / GPUMarker.h, into oracle/_ref/libvqref_mip.so. TEST INFRASTRUCTURE:
// pins vqo_mip_chain_min_rgba32f / vqo_mip_chain_box_rgba8 bit for bit (tests/test_ref_pinning.py). Never loaded by the product.
#include "Renderer/Resources/DXGIUtils.cpp"

extern "C" void vqref_mip_image(const void* src, void* dst, unsigned width, unsigned height, unsigned bytesPerPixel) {
    VQ_DXGI_UTILS::MipImage(src, dst, width, height, bytesPerPixel);
}
