//! rmqtt-gpu-router plugin: installs `GpuRouter` into `extends.router` at start(), the same way
//! rmqtt-cluster-broadcast/src/lib.rs:141-142 and rmqtt-cluster-raft/src/lib.rs:384 install theirs.
//! Source only — see Cargo.toml.
mod batcher;
mod ffi;
mod router;

use async_trait::async_trait;
use rmqtt::context::ServerContext;
use rmqtt::plugin::{PackageInfo, Plugin};
use rmqtt::register;
use rmqtt::router::DefaultRouter;
use rmqtt::Result;

pub use router::GpuRouter;

register!(GpuRouterPlugin::new);

struct GpuRouterPlugin {
    scx: ServerContext,
    router: GpuRouter,
}

impl GpuRouterPlugin {
    async fn new<S: Into<String>>(scx: ServerContext, _name: S) -> Result<Self> {
        let device = std::env::var("RMQTT_GPU_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let router = GpuRouter::new(DefaultRouter::new(Some(scx.clone())), device)?;
        Ok(Self { scx, router })
    }
}

#[async_trait]
impl Plugin for GpuRouterPlugin {
    async fn init(&mut self) -> Result<()> { Ok(()) }

    async fn start(&mut self) -> Result<()> {
        // rmqtt/src/extend.rs:135 — the router slot is an RwLock<Box<dyn Router>>
        *self.scx.extends.router_mut().await = Box::new(self.router.clone());
        log::info!("gpu router installed");
        Ok(())
    }

    async fn stop(&mut self) -> Result<bool> { Ok(false) } // like the cluster routers: not hot-removable
}

impl PackageInfo for GpuRouterPlugin {
    fn name(&self) -> &str { "rmqtt-gpu-router" }
}
