fn main() {
    // RMQTT_GPU_ROUTER_LIB_DIR = directory holding librmqtt_gpu_router.so
    if let Ok(dir) = std::env::var("RMQTT_GPU_ROUTER_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
    }
    println!("cargo:rustc-link-lib=dylib=rmqtt_gpu_router");
}
