//! rmqtt-gpu-router plugin: installs `GpuRouter` into `extends.router` and `GpuRetainStorage` into
//! `extends.retain` at start(), the same way rmqtt-cluster-broadcast/src/lib.rs:141-142,
//! rmqtt-cluster-raft/src/lib.rs:384 and rmqtt-retainer/src/lib.rs:191 install theirs.
//!
//! Configuration (environment; a plugin config file would carry the same keys):
//!   RMQTT_GPU_DEVICES        comma-separated HIP device ordinals, one table shard per device (default "0")
//!   RMQTT_GPU_MAX_BATCH      publishes per device pass at most (default 4096)
//!   RMQTT_GPU_MAX_DELAY_US   micro-batcher deadline (default 150)
//!   RMQTT_GPU_RETAIN         "1": also serve retained-message wildcard queries from the GPU (default off)
//! Source only — see Cargo.toml.
mod batcher;
mod ffi;
mod retain;
mod router;

use std::time::Duration;

use async_trait::async_trait;
use rmqtt::context::ServerContext;
use rmqtt::plugin::{PackageInfo, Plugin};
use rmqtt::register;
use rmqtt::Result;

pub use retain::{GpuRetainIndex, GpuRetainStorage};
pub use router::GpuRouter;

register!(GpuRouterPlugin::new);

struct GpuRouterPlugin {
    scx: ServerContext,
    router: GpuRouter,
    retain: Option<std::sync::Arc<GpuRetainStorage>>,
}

fn env_or<T: std::str::FromStr>(key: &str, default: T) -> T {
    std::env::var(key).ok().and_then(|s| s.parse().ok()).unwrap_or(default)
}

impl GpuRouterPlugin {
    async fn new<S: Into<String>>(scx: ServerContext, _name: S) -> Result<Self> {
        let devices: Vec<i32> = std::env::var("RMQTT_GPU_DEVICES")
            .unwrap_or_else(|_| "0".into())
            .split(',')
            .filter_map(|s| s.trim().parse().ok())
            .collect();
        let router = GpuRouter::new(
            scx.clone(),
            &devices,
            env_or("RMQTT_GPU_MAX_BATCH", 4096usize),
            Duration::from_micros(env_or("RMQTT_GPU_MAX_DELAY_US", 150u64)),
        )?;
        let retain = if env_or("RMQTT_GPU_RETAIN", 0u8) == 1 { Some(std::sync::Arc::new(GpuRetainStorage::new(devices[0])?)) } else { None };
        Ok(Self { scx, router, retain })
    }
}

#[async_trait]
impl Plugin for GpuRouterPlugin {
    async fn init(&mut self) -> Result<()> { Ok(()) }

    async fn start(&mut self) -> Result<()> {
        // rmqtt/src/extend.rs:135 — the router slot is an RwLock<Box<dyn Router>>
        *self.scx.extends.router_mut().await = Box::new(self.router.clone());
        if let Some(r) = &self.retain {
            // rmqtt/src/extend.rs:71 — the retain slot, as rmqtt-retainer/src/lib.rs:191 fills it
            *self.scx.extends.retain_mut().await = Box::new(RetainHandle(r.clone()));
        }
        log::info!("gpu router installed");
        Ok(())
    }

    async fn stop(&mut self) -> Result<bool> { Ok(false) } // like the cluster routers: not hot-removable
}

impl PackageInfo for GpuRouterPlugin {
    fn name(&self) -> &str { "rmqtt-gpu-router" }
}

/// `Box<dyn RetainStorage>` over the shared storage (the plugin keeps a handle for expiry sweeps).
struct RetainHandle(std::sync::Arc<GpuRetainStorage>);

#[async_trait]
impl rmqtt::retain::RetainStorage for RetainHandle {
    fn enable(&self) -> bool { true }
    async fn set(&self, topic: &rmqtt::types::TopicName, retain: rmqtt::types::Retain, expiry: Option<Duration>) -> Result<()> {
        self.0.set(topic, retain, expiry).await
    }
    async fn get(&self, f: &rmqtt::types::TopicFilter) -> Result<Vec<(rmqtt::types::TopicName, rmqtt::types::Retain)>> { self.0.get(f).await }
    async fn count(&self) -> isize { self.0.count().await }
    async fn max(&self) -> isize { self.0.max().await }
}
