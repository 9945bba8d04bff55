//! rmqtt-gpu-router plugin: installs `GpuRouter` into `extends.router` and `GpuRetainStorage` into
//! `extends.retain` at start(), the same way rmqtt-cluster-broadcast/src/lib.rs:141-142,
//! rmqtt-cluster-raft/src/lib.rs:384 and rmqtt-retainer/src/lib.rs:191 install theirs.
//!
//! Configuration: the plugin's own config file, read the way every rmqtt plugin reads its own
//! (`scx.plugins.read_config_default::<PluginConfig>(&name)`, rmqtt/src/plugin.rs:692, as rmqtt-retainer/src/lib.rs:64 does) —
//! `rmqtt-plugins/rmqtt-gpu-router.toml`:
//!   devices = [0]           HIP device ordinals, one table shard per device
//!   max_batch = 4096        publishes per device pass at most
//!   max_delay_us = 150      micro-batcher deadline
//!   retain = false          also serve retained-message wildcard queries from the GPU
//!   forwards = true         install `GpuShared` too: `Shared::forwards` consumes the device's delivery stage directly (no SubRelationsMap)
//! Environment variables of the same names in upper case with the `RMQTT_GPU_` prefix override the file (round 2 read only those).
//! Source only — see Cargo.toml.
mod batcher;
mod ffi;
mod message_index;
mod retain;
mod router;
mod shared;

use std::time::Duration;

use async_trait::async_trait;
use rmqtt::context::ServerContext;
use rmqtt::plugin::{PackageInfo, Plugin};
use rmqtt::register;
use rmqtt::Result;

pub use message_index::GpuMessageIndex;
pub use retain::{GpuRetainIndex, GpuRetainStorage};
pub use router::GpuRouter;
pub use shared::GpuShared;

register!(GpuRouterPlugin::new);

struct GpuRouterPlugin {
    scx: ServerContext,
    router: GpuRouter,
    shared: Option<GpuShared>,
    retain: Option<std::sync::Arc<GpuRetainStorage>>,
}

fn env_or<T: std::str::FromStr>(key: &str, default: T) -> T {
    std::env::var(key).ok().and_then(|s| s.parse().ok()).unwrap_or(default)
}

/// `rmqtt-plugins/rmqtt-gpu-router.toml` (every key optional)
#[derive(Debug, Clone, serde::Deserialize)]
#[serde(default)]
struct PluginConfig {
    devices: Vec<i32>,
    max_batch: usize,
    max_delay_us: u64,
    retain: bool,
    forwards: bool,
}
impl Default for PluginConfig {
    fn default() -> Self {
        Self { devices: vec![0], max_batch: 4096, max_delay_us: 150, retain: false, forwards: true }
    }
}

impl GpuRouterPlugin {
    async fn new<S: Into<String>>(scx: ServerContext, name: S) -> Result<Self> {
        let name = name.into();
        let mut cfg = scx.plugins.read_config_default::<PluginConfig>(&name)?;
        if let Ok(d) = std::env::var("RMQTT_GPU_DEVICES") {
            cfg.devices = d.split(',').filter_map(|s| s.trim().parse().ok()).collect();
        }
        if cfg.devices.is_empty() { cfg.devices = vec![0]; }
        cfg.max_batch = env_or("RMQTT_GPU_MAX_BATCH", cfg.max_batch);
        cfg.max_delay_us = env_or("RMQTT_GPU_MAX_DELAY_US", cfg.max_delay_us);
        cfg.retain = env_or("RMQTT_GPU_RETAIN", cfg.retain as u8) == 1;
        cfg.forwards = env_or("RMQTT_GPU_FORWARDS", cfg.forwards as u8) == 1;
        log::info!("{name} config: {cfg:?}");
        let router = GpuRouter::new(scx.clone(), &cfg.devices, cfg.max_batch, Duration::from_micros(cfg.max_delay_us))?;
        let retain = if cfg.retain { Some(std::sync::Arc::new(GpuRetainStorage::new(cfg.devices[0])?)) } else { None };
        let shared = if cfg.forwards { Some(GpuShared::new(scx.clone(), router.clone(), cfg.max_batch, Duration::from_micros(cfg.max_delay_us))) } else { None };
        Ok(Self { scx, router, shared, retain })
    }
}

#[async_trait]
impl Plugin for GpuRouterPlugin {
    async fn init(&mut self) -> Result<()> { Ok(()) }

    async fn start(&mut self) -> Result<()> {
        // rmqtt/src/extend.rs:135 — the router slot is an RwLock<Box<dyn Router>>
        *self.scx.extends.router_mut().await = Box::new(self.router.clone());
        if let Some(s) = &self.shared {
            // rmqtt/src/extend.rs:123 — the shared slot, replaced the way rmqtt-cluster-broadcast/src/lib.rs:141 replaces it
            *self.scx.extends.shared_mut().await = Box::new(s.clone());
        }
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
